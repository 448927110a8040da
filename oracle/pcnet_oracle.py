"""CPU restatement of ``PitchClassNet.forward`` (default architecture family).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Stock torch CPU ops,
any float dtype (float64 is the reference's dtype, ``models.py:199,237,739``).

Every function cites the reference lines it restates (``/root/reference``).
The restatement is purely functional: it consumes a reference-format
``state_dict`` (key names of SURVEY.md section 8b) and never builds modules, so
it shares no code path with the product's drop-in class.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.01   # nn.LeakyReLU() default, models.py:197,234,315
BN_EPS = 1e-5        # nn.BatchNorm2d default, models.py:196,233,314


_BN_SINK = None      # set by record_bn_stats(): list of (prefix, batch mean, biased batch var, elements per channel)


class record_bn_stats:
    """Context manager: collect the batch statistics of every train-mode BatchNorm the forward visits, so that a caller
    can restate nn.BatchNorm2d's side effect (``running <- 0.9*running + 0.1*(mean, unbiased var)``, torch defaults; the
    reference never overrides momentum, models.py:196,233,314)."""

    def __enter__(self):
        global _BN_SINK
        self.rows = _BN_SINK = []
        return self.rows

    def __exit__(self, *exc):
        global _BN_SINK
        _BN_SINK = None


def update_running_stats(sd, rows, momentum=0.1, backward_ran=True):
    """Apply torch's train-mode running-statistics update to ``sd`` in place for the rows of ``record_bn_stats``.

    ``backward_ran``: the reference checkpoints norm1 + conv1 of every dense layer (``cp.checkpoint``, models.py:484-489, 553), so a
    training step's BACKWARD runs those BatchNorms a second time in train mode: they blend the same batch statistics twice and count two
    batches per step (pinned by tests/golden/pcnet_denseblock_train_T40.npz)."""
    with torch.no_grad():
        for prefix, mean, var, count in rows:
            unbiased = var * count / max(count - 1, 1)
            for _ in range(2 if backward_ran and ".denselayer" in prefix and prefix.endswith(".norm1.") else 1):
                sd[prefix + "running_mean"].mul_(1 - momentum).add_(momentum * mean.detach().to(sd[prefix + "running_mean"].dtype))
                sd[prefix + "running_var"].mul_(1 - momentum).add_(momentum * unbiased.detach().to(sd[prefix + "running_var"].dtype))
                key = prefix + "num_batches_tracked"
                if key in sd:
                    sd[key] += 1


def _bn(x, sd, prefix, training=False, stats=None):
    """BatchNorm2d, eval mode = running stats (models.py:196 etc.).

    ``training=True`` restates train-mode normalisation (batch statistics,
    biased variance) as ``equivariance_test.py:178`` runs the net; running
    buffers are not updated here (see ``record_bn_stats`` / ``update_running_stats``).
    """
    w, b = sd[prefix + "weight"], sd[prefix + "bias"]
    if training:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        if _BN_SINK is not None:
            _BN_SINK.append((prefix, mean, var, x.numel() // x.shape[1]))
    else:
        mean, var = sd[prefix + "running_mean"], sd[prefix + "running_var"]
    if stats is not None:
        stats[prefix] = (mean, var)
    scale = w / torch.sqrt(var + BN_EPS)
    return (x - mean[None, :, None, None]) * scale[None, :, None, None] + b[None, :, None, None]


def _lrelu(x):
    return F.leaky_relu(x, LRELU_SLOPE)


def equiv_pc_conv(x, weight, bias, same: bool):
    """EquivariantPitchClassConvolutionSimple.forward, models.py:36-47.

    Wrap the first 11 pitch-class rows below the 12 (``x_wrap``, :45) and run a
    plain Conv2d with a (12, kd) kernel; zero 'same' padding in time only (:28).
    """
    pcs = weight.shape[2]
    assert x.shape[2] == pcs                                   # models.py:44
    x_wrap = torch.cat([x, x[:, :, 0:pcs - 1, :]], dim=2)      # models.py:45
    kd = weight.shape[3]
    return F.conv2d(x_wrap, weight, bias, padding=(0, kd // 2 if same else 0))


def pitch2pitchclass_pool(x, pitch_classes: int = 12):
    """Pitch2PitchClassPool.forward, models.py:95-106 (ctor :84-92).

    Dilated max-pool over octaves; -inf rows are appended only when the row
    count is not a multiple of 12.
    """
    rows = x.shape[2]
    ks = math.ceil(rows / pitch_classes)
    pad = ks * pitch_classes - rows
    if pad:
        filler = torch.full((x.shape[0], x.shape[1], pad, x.shape[3]), float("-inf"), dtype=x.dtype)
        x = torch.cat([x, filler], dim=2)
    return F.max_pool2d(x, (ks, 1), (1, 1), dilation=(pitch_classes, 1))


def pitch2pitchclass_conv(x, sd, prefix, training=False):
    """Pitch2PitchClassConv.forward, models.py:108-133 (--p2pc_conv): the octave fold as a learned convolution -- kernel
    (ceil(pitches_in / 12), 1) with dilation (12, 1) over the channels, then BatchNorm + LeakyReLU -- instead of the max."""
    w = sd[prefix + "conv.weight"]
    pad = w.shape[2] * 12 - x.shape[2]                       # rows of padding_value appended (:130-131); 0 for whole octaves
    if pad:
        x = torch.cat([x, torch.full((x.shape[0], x.shape[1], pad, x.shape[3]), float("-inf"), dtype=x.dtype)], dim=2)
    y = F.conv2d(x, w, sd[prefix + "conv.bias"], dilation=(12, 1))
    return _lrelu(_bn(y, sd, prefix + "bn.", training))


def pitchclass2pitch(x, target_rows: int):
    """PitchClass2Pitch.forward, models.py:140-143: tile rows, crop."""
    reps = math.ceil(target_rows / x.shape[2])
    return x.repeat(1, 1, reps, 1)[:, :, 0:target_rows, :]


def _circular_conv(x, weight, bias, stride, pad_hw):
    """nn.Conv2d(..., padding=pad_hw, padding_mode='circular') (models.py:230,313)."""
    ph, pw = pad_hw
    x = F.pad(x, (pw, pw, ph, ph), mode="circular")
    return F.conv2d(x, weight, bias, stride=stride)


def _count(sd, prefix, suffix):
    """Number of conv blocks in an nn.Sequential(conv, bn, act, conv, bn, act, ...)."""
    n = 0
    while f"{prefix}{3 * n}{suffix}" in sd:
        n += 1
    return n


def _res_blocks(x, sd, prefix, conv, training, taps):
    """--resblock (models.py:181-187 / :218-224): after the first conv + BN + LeakyReLU the Sequential holds ``conv_layers`` blocks
    at indices 3, 4, ...; a block is x -> act2(x + b2(conv2(act1(b1(conv1(x)))))) (ResBlock / ResBlockEquivariant, models.py:402-454).
    ``conv(x, block_prefix + "conv1")`` applies the stack's convolution type."""
    idx = 3
    while f"{prefix}layer.{idx}.b1.weight" in sd:
        bp = f"{prefix}layer.{idx}."
        h = _lrelu(_bn(conv(x, bp + "conv1"), sd, bp + "b1.", training))
        x = _lrelu(x + _bn(conv(h, bp + "conv2"), sd, bp + "b2.", training))
        if taps is not None:
            taps[f"{prefix}layer.{idx}"] = x
        idx += 1
    return x


def _dense_block(x, sd, prefix, conv, training, taps):
    """--denseblock (DenseBlock / DenseBlockEquivariant, models.py:584-648; layers _DenseLayer / _DenseLayerEquivariant, :456-582):
    every layer reads the concatenation of the block input and all earlier layers' outputs,
        new = conv2(relu(norm2(conv1(leaky_relu(norm1(cat))))))          (:473-476, 516-517 / :536-539, 579-580)
    (pre-activation BatchNorm; relu1 is a LeakyReLU, relu2 a plain ReLU; conv1 is the 1-wide bottleneck), and the block returns the
    concatenation of everything (:614 / :647).  drop_rate is 0.0 at both call sites (:189, 226).  ``conv(x, key_prefix, same)``
    applies the stack's convolution type."""
    feats = [x]
    j = 1
    while f"{prefix}denselayer{j}.norm1.weight" in sd:
        lp = f"{prefix}denselayer{j}."
        cat = torch.cat(feats, dim=1)
        h = conv(_lrelu(_bn(cat, sd, lp + "norm1.", training)), lp + "conv1")
        new = conv(F.relu(_bn(h, sd, lp + "norm2.", training)), lp + "conv2")
        feats.append(new)
        if taps is not None:
            taps[f"{prefix}denselayer{j}"] = new
        j += 1
    return torch.cat(feats, dim=1)


def pc2pc_stack(pc, sd, prefix, training=False, taps=None):
    """PitchClass2PitchClass default branch, models.py:190-197, 201-203; --resblock branch :181-187; --denseblock branch :188-189."""
    if f"{prefix}layer.0.denselayer1.norm1.weight" in sd:
        # conv1: EquivariantPitchClassConvolutionSimple with kernel_depth 1 (12 x 1, no time padding needed), conv2: 12 x k, same padding
        return _dense_block(pc, sd, prefix + "layer.0.", lambda x, q: equiv_pc_conv(x, sd[q + ".conv2d.weight"], sd[q + ".conv2d.bias"], same=True),
                            training, taps)
    if f"{prefix}layer.3.b1.weight" in sd:
        pc = equiv_pc_conv(pc, sd[prefix + "layer.0.conv2d.weight"], sd[prefix + "layer.0.conv2d.bias"], same=True)
        pc = _lrelu(_bn(pc, sd, prefix + "layer.1.", training))
        return _res_blocks(pc, sd, prefix, lambda x, q: equiv_pc_conv(x, sd[q + ".conv2d.weight"], sd[q + ".conv2d.bias"], same=True),
                           training, taps)
    n = _count(sd, prefix + "layer.", ".conv2d.weight")
    for i in range(n):
        pc = equiv_pc_conv(pc, sd[f"{prefix}layer.{3*i}.conv2d.weight"], sd[f"{prefix}layer.{3*i}.conv2d.bias"], same=True)
        pc = _lrelu(_bn(pc, sd, f"{prefix}layer.{3*i+1}.", training))
        if taps is not None:
            taps[f"{prefix}layer.{3*i+2}"] = pc
    return pc


def p2p_stack(p, sd, prefix, training=False, taps=None):
    """Pitch2Pitch default branch, models.py:227-234, 239-243: circular on both axes; --resblock branch :218-224; --denseblock
    branch :225-226 (plain nn.Conv2d without bias: 1 x 1, then k x k with ZERO padding k // 2 on both axes, :464, 468)."""
    if f"{prefix}layer.0.denselayer1.norm1.weight" in sd:
        return _dense_block(p, sd, prefix + "layer.0.", lambda x, q: F.conv2d(x, sd[q + ".weight"], None, padding=sd[q + ".weight"].shape[2] // 2),
                            training, taps)
    if f"{prefix}layer.3.b1.weight" in sd:
        circ = lambda x, q: _circular_conv(x, sd[q + ".weight"], sd[q + ".bias"], (1, 1), (sd[q + ".weight"].shape[2] // 2,) * 2)
        p = _lrelu(_bn(circ(p, prefix + "layer.0"), sd, prefix + "layer.1.", training))
        return _res_blocks(p, sd, prefix, circ, training, taps)
    n = _count(sd, prefix + "layer.", ".weight")
    for i in range(n):
        w = sd[f"{prefix}layer.{3*i}.weight"]
        k = w.shape[2]
        p = _circular_conv(p, w, sd[f"{prefix}layer.{3*i}.bias"], (1, 1), (k // 2, k // 2))
        p = _lrelu(_bn(p, sd, f"{prefix}layer.{3*i+1}.", training))
        if taps is not None:
            taps[f"{prefix}layer.{3*i+2}"] = p
    return p


def semitone_pool(p, sd, prefix, training=False):
    """pool_semi + BN + LeakyReLU, models.py:313-315 / :337-339, used :361-363, :386-388.

    3x3 conv, stride (3,1), circular padding (0,1): three third-semitone bins ->
    one semitone, time wraps.
    """
    x = _circular_conv(p, sd[prefix + "pool_semi.weight"], sd[prefix + "pool_semi.bias"], (3, 1), (0, 1))
    return _lrelu(_bn(x, sd, prefix + "pool_semi_b.", training))


def pitchclass2pitch_memory(p, p_sixth):
    """PitchClass2Pitch_MemoryVariant.forward, models.py:145-166 (--pc2p_mem): instead of concatenating the repeated third-semitone
    map, ADD it to the pitch stream -- after summing groups of its channels down to the stream's channel count.  The reference
    reshapes the P pitch rows to (36, P / 36), so row r receives third-semitone index r // (P / 36) (eight CONSECUTIVE rows share
    one), not r % 36 as the repeat of the default path does; kept as it is."""
    B, C, P, T = p.shape
    s = p_sixth.reshape(B, C, p_sixth.shape[1] // C, p_sixth.shape[2], T).sum(dim=2)          # (B, C, 36, T)
    k = s.shape[2]
    return (p.reshape(B, C, k, P // k, T) + s.reshape(B, C, k, 1, T)).reshape(B, C, P, T)


def forward_features(sd, mel, time_pool_size=2, training=False, taps=None):
    """nn.Sequential of PitchClassNetLayer.forward, models.py:352-399 (default flags; --pc2p_mem when the first pitch conv of a
    layer takes only the pitch stream's channels)."""
    num_layers = 0
    while (f"model.{num_layers}.pc2pc.layer.0.conv2d.weight" in sd       # (pool_semi is absent from --stay_sixth layers >= 1)
           or f"model.{num_layers}.pc2pc.layer.0.denselayer1.norm1.weight" in sd):
        num_layers += 1
    p, pc = mel, None
    pitches = mel.shape[2]
    # --stay_sixth (models.py:322-323, 336, 366-367, 371, 385): the pitch stream continues at semitone resolution -- layer 0's semitone
    # map replaces the CQT as the stream, later layers have neither up_sixth nor pool_semi and repeat the 12 pitch-class rows directly
    stay = num_layers > 1 and "model.1.up_sixth.weight" not in sd
    for i in range(num_layers):
        pre = f"model.{i}."
        if i == 0:
            p_semi = semitone_pool(p, sd, pre, training)            # :361-363
            if stay:
                p = p_semi                                          # :366-367
            pc = (pitch2pitchclass_conv(p_semi, sd, pre + "pool.", training) if pre + "pool.conv.weight" in sd
                  else pitch2pitchclass_pool(p_semi))               # :368 (p stays raw, :366-367)
            if taps is not None:
                taps[pre + "pool"] = pc
            pc = pc2pc_stack(pc, sd, pre + "pc2pc.", training, taps)  # :369
        elif stay:
            p = torch.cat([p, pitchclass2pitch(pc, p.shape[2])], dim=1)   # :379-383 with up = PitchClass2Pitch(pitches // 3)
            p = p2p_stack(p, sd, pre + "p2p.", training, taps)      # :384
            pc2 = (pitch2pitchclass_conv(p, sd, pre + "pool.", training) if pre + "pool.conv.weight" in sd
                   else pitch2pitchclass_pool(p))                   # :391
            if taps is not None:
                taps[pre + "pool"] = pc2
            pc = torch.cat([pc, pc2], dim=1)                        # :392
            pc = pc2pc_stack(pc, sd, pre + "pc2pc.", training, taps)  # :393
            p = F.max_pool2d(p, (1, time_pool_size))                # :395
            pc = F.max_pool2d(pc, (1, time_pool_size))              # :396
            if taps is not None:
                taps[pre + "time_pool_pc"] = pc
        else:
            p_sixth = F.conv_transpose2d(pc, sd[pre + "up_sixth.weight"], sd[pre + "up_sixth.bias"], stride=(3, 1))  # :372
            p_sixth = _lrelu(_bn(p_sixth, sd, pre + "up_sixth_b.", training))     # :373-374
            if taps is not None:
                taps[pre + "up_sixth_a"] = p_sixth
            if pre + "p2p.layer.0.weight" in sd and sd[pre + "p2p.layer.0.weight"].shape[1] == p.shape[1]:       # --pc2p_mem: :376-377, no concat (:382)
                p = pitchclass2pitch_memory(p, p_sixth)
            else:
                p2 = pitchclass2pitch(p_sixth, pitches)             # :378
                p = torch.cat([p, p2], dim=1)                       # :383
            p = p2p_stack(p, sd, pre + "p2p.", training, taps)      # :384
            pc2 = semitone_pool(p, sd, pre, training)               # :386-388
            pc2 = (pitch2pitchclass_conv(pc2, sd, pre + "pool.", training) if pre + "pool.conv.weight" in sd
                   else pitch2pitchclass_pool(pc2))                 # :389
            if taps is not None:
                taps[pre + "pool"] = pc2
            pc = torch.cat([pc, pc2], dim=1)                        # :392
            pc = pc2pc_stack(pc, sd, pre + "pc2pc.", training, taps)  # :393
            p = F.max_pool2d(p, (1, time_pool_size))                # :395
            pc = F.max_pool2d(pc, (1, time_pool_size))              # :396
            if taps is not None:
                taps[pre + "time_pool_pc"] = pc
    return p, pc, num_layers


def _equiv_head(pc, sd, name, training=False):
    """tonic/key classifier Sequential, models.py:716-731, applied :750-751."""
    n_hidden = _count(sd, name + ".", ".conv2d.weight")
    # hidden blocks sit at indices 0,3,6..; the last conv (no BN) closes the Sequential
    idx = 0
    x = pc
    while f"{name}.{idx}.conv2d.weight" in sd:
        x = equiv_pc_conv(x, sd[f"{name}.{idx}.conv2d.weight"], sd[f"{name}.{idx}.conv2d.bias"], same=False)
        if f"{name}.{idx+1}.weight" in sd:       # BN follows -> hidden block
            x = _lrelu(_bn(x, sd, f"{name}.{idx+1}.", training))
            idx += 3
        else:
            break
    return x


def _genre_head(pc, sd, training=False):
    """genre classifier: plain Conv2d (1,k) [+BN+LReLU] ..., Conv2d (2,k); models.py:724,733."""
    idx = 0
    x = pc
    while f"genre_classifier.{idx}.weight" in sd and sd[f"genre_classifier.{idx}.weight"].dim() == 4:
        x = F.conv2d(x, sd[f"genre_classifier.{idx}.weight"], sd[f"genre_classifier.{idx}.bias"])
        if f"genre_classifier.{idx+1}.running_mean" in sd:
            x = _lrelu(_bn(x, sd, f"genre_classifier.{idx+1}.", training))
            idx += 3
        else:
            break
    return x


def pcnet_forward(sd: Dict[str, torch.Tensor], mel: torch.Tensor, seq_length: Optional[torch.Tensor],
                  kernel_size: int = 7, head_layers: int = 2, time_pool_size: int = 2,
                  genre: Optional[bool] = None, max_pool: bool = False, training: bool = False,
                  taps: Optional[dict] = None, local_window: Optional[int] = None) -> Tuple[torch.Tensor, ...]:
    """PitchClassNet.forward, models.py:747-817.

    Returns ``(key_out, tonic_out[, genre_out])`` exactly as the reference:
    sigmoid on key only (:802), 2-tuple when there is no genre head (:815).

    ``local_window`` = ``opt.frames * opt.loc_window_size - head_layers * (kernel_size - 1)`` selects ``--local``
    (models.py:720-722, 805-810): the key / tonic heads end in ``MaxPool2d((1, W), stride=1)`` and the maps are returned per
    frame -- *reshaped*, not transposed, to ``(B, T', 12)`` (the reference's ``reshape``, kept as it is), genre ``(B, Tm, 11)``.
    """
    if genre is None:
        genre = "genre_classifier.0.weight" in sd
    if local_window is not None:
        time_pool_size = 1                                                           # :348, :394 -- no time pooling with --local
    p, pc, num_layers = forward_features(sd, mel, time_pool_size, training, taps)   # :749
    tonic = _equiv_head(pc, sd, "tonic_classifier", training)                        # :750
    key = _equiv_head(pc, sd, "key_classifier", training)                            # :751
    gen = _genre_head(pc, sd, training) if genre else None                           # :753
    if taps is not None:
        taps["tonic_map"], taps["key_map"] = tonic, key
        if genre:
            taps["genre_map"] = gen
    if local_window is not None:                                                     # :720-722, :805-810
        tonic = F.max_pool2d(tonic, kernel_size=(1, local_window), stride=1)
        key = F.max_pool2d(key, kernel_size=(1, local_window), stride=1)
        tonic_out = tonic.reshape(tonic.shape[0], tonic.shape[3], tonic.shape[2])
        key_out = torch.sigmoid(key.reshape(key.shape[0], key.shape[3], key.shape[2]))
        if genre:
            return key_out, tonic_out, gen.reshape(gen.shape[0], gen.shape[3], gen.shape[2])
        return key_out, tonic_out

    def pool_all(x):
        return x.max(dim=-1).values if max_pool else x.mean(dim=-1)

    if seq_length is not None:
        # :757-760 -- floor per layer, then subtract the heads' valid-conv shrink
        actual = seq_length.reshape(-1).to(torch.float64)
        for _ in range(num_layers - 1):
            actual = torch.floor(actual / time_pool_size)
        actual = actual.to(torch.int32) - (kernel_size - 1) * head_layers
        if actual.numel() == 1 and tonic.shape[0] > 1:
            actual = actual.expand(tonic.shape[0])

        def pool_masked(x):
            rows = []
            for j in range(x.shape[0]):
                seg = x[j, :, :, : int(actual[j])]
                # :764-785 -- quirk kept: with max_pool only sample 0 takes the max
                if max_pool and j == 0:
                    rows.append(seg.max(dim=-1).values)
                else:
                    rows.append(seg.mean(dim=-1))
            return torch.stack(rows, 0)

        tonic_out, key_out = pool_masked(tonic), pool_masked(key)
        genre_out = pool_masked(gen) if genre else None
    else:                                                                            # :786-797
        tonic_out, key_out = pool_all(tonic), pool_all(key)
        genre_out = pool_all(gen) if genre else None

    tonic_out = tonic_out.flatten(1)                                                 # :800
    key_out = torch.sigmoid(key_out.flatten(1))                                      # :801-802
    if genre:
        return key_out, tonic_out, genre_out.flatten(1)                              # :813
    return key_out, tonic_out                                                        # :815


def to_dtype(sd, dtype):
    """Cast the float entries of a state_dict (``num_batches_tracked`` stays int64)."""
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
