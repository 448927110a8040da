"""PARITY (GPU): the training step around the HIP forward/backward -- device-resident weight repacking, the fused Adam
kernel, accumulate_grad_batches, the loss curve of a short fit against the float64 oracle loop (SURVEY.md section 8d,
config 3), and the 2-rank data-parallel step (config 4's mechanics on one GPU, gloo staging instead of RCCL)."""
import copy
import ctypes as C
import json
import os
import socket
from argparse import Namespace

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import ake_amd
from ake_amd import _lib
from conftest import golden_state_dict, load_golden
from oracle import loss_oracle, pcnet_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def default_net(gold):
    opt = Namespace(**json.loads(str(gold["opt"])))
    net = ake_amd.PitchClassNet(288, 12, 2, 7, opt)
    net.load_state_dict(golden_state_dict(gold), strict=True)
    return net, opt


def make_batch(batch, frames, seed, genre_classes=11):
    g = torch.Generator().manual_seed(seed)
    mel = torch.rand((batch, 1, 288, frames), generator=g) * 2.5
    seq = torch.randint(frames - 12, frames + 1, (batch,), generator=g)
    key = torch.randint(0, 24, (batch,), generator=g)
    rows = torch.from_numpy(np.asarray(ake_amd.KEY_SIGNATURE_MAP, dtype=np.float32))
    key_labels = rows[key % rows.shape[0]]
    tonic = torch.nn.functional.one_hot(key % 12, 12).float()
    sig = torch.nn.functional.one_hot(key, 24).float()
    genre = torch.nn.functional.one_hot(torch.randint(0, genre_classes, (batch,), generator=g), genre_classes).float()
    genre[0] = 0                                                     # one clip without a genre label (models.py:839 mask)
    return {"mel": mel, "key_labels": key_labels, "tonic_labels": tonic, "key_signature_id": sig, "genre": genre, "seq_length": seq}


# ---------------------------------------------------------------------------------------------- weights on the device
def test_device_repack_is_bit_identical_to_host_finalize(gold_default):
    """ake_pcnet_load_from_device_f32 (index-map gather + on-device BatchNorm fold) == set_tensor + finalize."""
    L = _lib.lib()
    sd = golden_state_dict(gold_default)
    cfg = _lib.PcnetConfig()
    _lib.check(L.ake_pcnet_default_config(C.byref(cfg), 8, 1), "cfg")
    handles = []
    for _ in range(2):
        h = C.c_void_p()
        _lib.check(L.ake_pcnet_create(C.byref(cfg), C.byref(h)), "create")
        handles.append(h)
    host_h, dev_h = handles
    total = int(L.ake_pcnet_grad_floats(dev_h))
    assert total == sum(v.numel() for v in sd.values() if v.is_floating_point())
    flat = torch.zeros(total, dtype=torch.float32)
    g = torch.Generator().manual_seed(1)
    for k, v in sd.items():
        if not v.is_floating_point():
            continue
        v = v.clone().float()
        if k.endswith("running_mean"):
            v = torch.randn(v.shape, generator=g) * 0.3          # make the eval fold non-trivial
        if k.endswith("running_var"):
            v = torch.rand(v.shape, generator=g) + 0.5
        shape = (C.c_int64 * max(1, v.dim()))(*v.shape)
        _lib.check(L.ake_pcnet_set_tensor(host_h, k.encode(), v.contiguous().data_ptr(), shape, v.dim()), k)
        off = int(L.ake_pcnet_grad_offset(dev_h, k.encode()))
        flat[off:off + v.numel()] = v.reshape(-1)
    _lib.check(L.ake_pcnet_finalize(host_h), "finalize")
    flat = flat.to(DEV)
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(L.ake_pcnet_load_from_device_f32(dev_h, flat.data_ptr(), stream), "load_from_device")
    x = torch.from_numpy(gold_default["x"]).to(DEV)
    seq = torch.from_numpy(gold_default["seq_length"]).to(DEV)
    B, _, _, T = x.shape
    outs = []
    for h in (host_h, dev_h):
        res = []
        for train in (False, True):
            key, tonic, genre = (torch.empty((B, n), device=DEV) for n in (12, 12, 11))
            if train:
                ws = torch.empty(int(L.ake_pcnet_train_workspace_bytes(h, B, T)), dtype=torch.uint8, device=DEV)
                _lib.check(L.ake_pcnet_forward_train_f32(h, x.data_ptr(), B, T, seq.data_ptr(), key.data_ptr(), tonic.data_ptr(), genre.data_ptr(),
                                                         None, ws.data_ptr(), ws.numel(), stream), "fwd train")
            else:
                ws = torch.empty(int(L.ake_pcnet_workspace_bytes(h, B, T)), dtype=torch.uint8, device=DEV)
                _lib.check(L.ake_pcnet_forward_f32(h, x.data_ptr(), B, T, seq.data_ptr(), key.data_ptr(), tonic.data_ptr(), genre.data_ptr(),
                                                   ws.data_ptr(), ws.numel(), stream), "fwd")
            res += [key.cpu(), tonic.cpu(), genre.cpu()]
        outs.append(res)
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    for h in handles:
        L.ake_pcnet_destroy(h)


def test_parameters_become_views_of_one_flat_buffer(gold_default):
    net, _ = default_net(gold_default)
    net = net.to(DEV).eval()
    x = torch.from_numpy(gold_default["x"]).to(DEV)
    seq = torch.from_numpy(gold_default["seq_length"]).to(DEV)
    ref = [t.clone() for t in net(x, seq)]
    flat, _ = net.flat_parameters()
    base = flat.data_ptr()
    for name, off, cnt in net._layout():
        assert dict(net.state_dict(keep_vars=True))[name].data_ptr() == base + 4 * off
    # the state_dict still round-trips (eval.py:115) and an in-place change through a parameter reaches the kernels
    sd = copy.deepcopy(net.state_dict())
    with torch.no_grad():
        net.key_classifier[3].conv2d.bias.add_(1.0)
    assert (net(x, seq)[0] - ref[0]).abs().max() > 1e-3
    net.load_state_dict(sd, strict=True)
    for a, b in zip(net(x, seq), ref):
        assert torch.equal(a, b)
    # float64 modules (train_model.py:106 .double()) keep their own storage and still run
    net64 = copy.deepcopy(net).double()
    for a, b in zip(net64(x.double(), seq), ref):
        assert a.dtype == torch.float64 and (a.float() - b).abs().max() < 1e-6


# ---------------------------------------------------------------------------------------------------------- optimizer
@pytest.mark.parametrize("weight_decay", [0.0, 0.01])
def test_fused_adam_kernel_matches_torch_adam(weight_decay):
    n = 100_003
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(n, generator=g)
    ref_p = p0.clone().requires_grad_(True)
    ref = torch.optim.Adam([ref_p], lr=3e-4, betas=(0.9, 0.999), weight_decay=weight_decay)      # models.py:1019
    sched = torch.optim.lr_scheduler.ExponentialLR(ref, gamma=0.96)
    p = p0.clone().to(DEV)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    mask = torch.ones(n, dtype=torch.uint8)
    mask[1000:2000] = 0
    mask_d = mask.to(DEV)
    L = _lib.lib()
    for step in range(1, 8):
        grad = torch.randn(n, generator=g) * (10.0 ** float(torch.randint(-4, 1, (1,), generator=g)))
        ref_p.grad = grad.clone()
        ref.step()
        lr = ref.param_groups[0]["lr"]
        gd = grad.to(DEV)
        _lib.check(L.ake_adam_step_f32(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), mask_d.data_ptr(), n, lr, 0.9, 0.999, 1e-8,
                                       weight_decay, step, 1.0, torch.cuda.current_stream().cuda_stream), "adam")
        if step % 3 == 0:
            sched.step()
        got = p.cpu()
        live = mask.bool()
        assert torch.equal(got[~live], p0[~live])                              # frozen entries untouched
        err = (got[live] - ref_p.detach()[live]).abs().max()
        assert err < 2e-7 * max(1.0, float(ref_p.detach().abs().max())), (step, float(err))


def test_accumulated_gradients_equal_the_sum(gold_default):
    net, _ = default_net(gold_default)
    net = net.to(DEV).train()
    b1, b2 = make_batch(3, 40, 1), make_batch(3, 40, 2)
    to = lambda b: {k: v.to(DEV) for k, v in b.items()}
    net.zero_grad()
    net.training_step(to(b1), 0)["loss"].backward()
    g1 = net._flat_grad.clone()
    net.zero_grad(set_to_none=True)
    net.training_step(to(b2), 0)["loss"].backward()
    g2 = net._flat_grad.clone()
    net.zero_grad(set_to_none=True)
    net.training_step(to(b1), 0)["loss"].backward()
    net.training_step(to(b2), 1)["loss"].backward()
    both = net._flat_grad
    assert (both - (g1 + g2)).abs().max() <= 2e-6 * (g1 + g2).abs().max()
    # and p.grad are the views the optimizer / clip_grad_norm_ see
    w = net.model[1].p2p.layer[0].weight
    off = net._grad_offsets()["model.1.p2p.layer.0.weight"]
    assert w.grad.data_ptr() == both.data_ptr() + 4 * off


def torch_loss(out, b):
    """models.py:878-893 with autograd (the numpy loss_oracle pins its value below)."""
    F = torch.nn.functional
    loss = F.binary_cross_entropy(out[0], b["key_labels"].to(out[0].dtype)) + F.cross_entropy(out[1], b["tonic_labels"].argmax(1))
    gl = b["genre"].long()
    mask = gl.sum(1) == 1
    if mask.sum() != 0:
        loss = loss + 0.1 * F.cross_entropy(out[2][mask], gl.argmax(1)[mask])
    return loss


def oracle_fit(sd32, opt, batches, acc, steps, lr=3e-4, gamma=0.96, dtype=torch.float64):
    """float64 restatement of Trainer.fit on the reference module: oracle forward (train-mode BN + torch's running-statistics update), general_step loss (loss_oracle), torch.optim.Adam + ExponentialLR.
    dtype=torch.float32: the same loop as stock float32 PyTorch on the CPU would run it (the band a float32 run occupies around the float64 curve)."""
    sd = {k: (v.to(dtype).clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and "num_batches" not in k
              else v.to(dtype).clone() if v.is_floating_point() else v.clone()) for k, v in sd32.items()}
    params = [v for v in sd.values() if torch.is_tensor(v) and v.requires_grad]
    optim = torch.optim.Adam(params, lr=lr, betas=(0.9, 0.999))
    losses = []
    done = 0
    for i, b in enumerate(batches):
        with pcnet_oracle.record_bn_stats() as rows:
            out = pcnet_oracle.pcnet_forward(sd, b["mel"].to(dtype), b["seq_length"], training=True)
        pcnet_oracle.update_running_stats(sd, rows)
        loss = torch_loss(out, b)
        if i == 0 and dtype == torch.float64:
            pinned = loss_oracle.general_step_loss(out[0].detach().numpy(), out[1].detach().numpy(), out[2].detach().numpy(),
                                                   b["key_labels"].numpy(), b["tonic_labels"].numpy(), b["genre"].numpy())
            assert abs(float(loss.detach()) - float(pinned)) < 1e-12
        (loss / acc).backward()
        losses.append(float(loss.detach()))
        if (i + 1) % acc == 0 or i + 1 == len(batches):
            optim.step(); optim.zero_grad(); done += 1
            if done >= steps:
                break
    return losses, sd


def test_short_fit_follows_the_oracle_loss_curve(gold_default):
    """Config 3's check in small: same batches, batch_size 4, accumulate 2, Adam lr 3e-4; per-batch training loss against
    the float64 loop.

    Tolerance: before the first optimizer step the losses agree to 1e-6.  Afterwards Adam's early steps move EVERY
    parameter by ~lr whatever its gradient's size, so the sign of noise-level gradient entries decides a +-lr move and
    float32 and float64 runs drift apart: float32 PyTorch on the CPU, run through the same oracle loop, is 2.8e-4 off
    the float64 curve right after step 1 and up to 4.3e-3 within these 16 batches (measured; DESIGN.md "Training
    parity").  The 1e-3 bar of SURVEY.md is therefore not attainable by any float32 implementation; the device has to
    stay within 1e-2, i.e. inside the band float32 PyTorch itself occupies."""
    net, opt = default_net(gold_default)
    sd32 = golden_state_dict(gold_default)
    batches = [make_batch(4, 40, 100 + i) for i in range(16)]
    acc, steps = 2, 8
    ref_losses, ref_sd = oracle_fit(sd32, opt, batches, acc, steps)
    net = net.to(DEV)
    trainer = ake_amd.Trainer(max_epochs=1, accumulate_grad_batches=acc)
    trainer.fit(net, train_dataloaders=batches, max_steps=steps)
    got = trainer.train_losses
    assert len(got) == len(ref_losses) == acc * steps
    rel = [abs(a - b) / abs(b) for a, b in zip(got, ref_losses)]
    print("loss curve (device / oracle):", [f"{a:.5f}/{b:.5f}" for a, b in zip(got, ref_losses)])
    assert max(rel[:acc]) < 1e-5 and max(rel) < 1e-2, rel
    assert got[-1] < got[0]                                        # and it is learning
    # weights after 8 Adam steps: every step moves a weight by ~lr, so agreement is relative to that scale
    dev_sd = net.state_dict()
    diffs = torch.cat([(dev_sd[k].cpu().double() - v.detach()).abs().reshape(-1) for k, v in ref_sd.items()
                       if v.is_floating_point() and "running" not in k])
    print("weights after", steps, "steps: max |dev - oracle| =", float(diffs.max()), " mean =", float(diffs.mean()), " (lr*steps =", steps * 3e-4, ")")
    assert float(diffs.max()) <= 2.05 * steps * 3e-4 and float(diffs.mean()) < 0.2 * steps * 3e-4
    for k, v in ref_sd.items():                                   # running statistics: momentum blend of batch statistics
        if "running" in k:
            assert float((dev_sd[k].cpu().double() - v).abs().max()) <= 2e-2 * max(1.0, float(v.abs().max())), k
    assert int(dev_sd["model.0.pool_semi_b.num_batches_tracked"]) == acc * steps


def test_resblock_short_fit_follows_the_oracle_loss_curve(gold_resblock):
    """The reference fixture's --resblock net (models.py:402-454; its weights, conv_layers = 3) through the same short fit: per-batch
    loss against the float64 loop, the blocks' BatchNorms (b1 / b2) collect running statistics, eval-mode inference still runs after."""
    net, opt = default_net(gold_resblock)
    assert net.resblock
    sd32 = golden_state_dict(gold_resblock)
    batches = [make_batch(4, 40, 300 + i) for i in range(8)]
    acc, steps = 2, 4
    ref_losses, ref_sd = oracle_fit(sd32, opt, batches, acc, steps)
    net = net.to(DEV)
    trainer = ake_amd.Trainer(max_epochs=1, accumulate_grad_batches=acc)
    trainer.fit(net, train_dataloaders=batches, max_steps=steps)
    got = trainer.train_losses
    assert len(got) == len(ref_losses) == acc * steps
    rel = [abs(a - b) / abs(b) for a, b in zip(got, ref_losses)]
    print("resblock loss curve (device / oracle):", [f"{a:.5f}/{b:.5f}" for a, b in zip(got, ref_losses)])
    assert max(rel[:acc]) < 1e-5 and max(rel) < 1e-2, rel
    dev_sd = net.state_dict()
    diffs = torch.cat([(dev_sd[k].cpu().double() - v.detach()).abs().reshape(-1) for k, v in ref_sd.items()
                       if v.is_floating_point() and "running" not in k])
    assert float(diffs.max()) <= 2.05 * steps * 3e-4 and float(diffs.mean()) < 0.2 * steps * 3e-4
    for k, v in ref_sd.items():
        if "running" in k:
            assert float((dev_sd[k].cpu().double() - v).abs().max()) <= 2e-2 * max(1.0, float(v.abs().max())), k
    assert int(dev_sd["model.1.p2p.layer.3.b2.num_batches_tracked"]) == acc * steps
    net.eval()
    with torch.no_grad():
        key, tonic, genre = net(batches[0]["mel"].to(DEV), batches[0]["seq_length"].to(DEV))
    ref = pcnet_oracle.pcnet_forward({k: v.detach() for k, v in ref_sd.items()}, batches[0]["mel"].double(), batches[0]["seq_length"])
    assert float((key.cpu().double() - ref[0]).abs().max()) < 2e-2 and torch.isfinite(tonic).all() and torch.isfinite(genre).all()


def test_denseblock_short_fit_follows_the_oracle_loss_curve(gold_denseblock):
    """The reference fixture's --denseblock net (models.py:456-648; its weights, n_filters = 2, conv_layers = 2) through the same short fit:
    per-batch loss against the float64 loop; every dense layer's norm1 counts TWO batches per step (the reference's checkpointed half runs
    again in backward) and its running statistics follow the double blend; eval-mode inference still runs after."""
    net, opt = default_net(gold_denseblock)
    assert net.denseblock
    sd32 = golden_state_dict(gold_denseblock)
    batches = [make_batch(4, 40, 500 + i) for i in range(8)]
    acc, steps = 2, 4
    ref_losses, ref_sd = oracle_fit(sd32, opt, batches, acc, steps)
    net = net.to(DEV)
    trainer = ake_amd.Trainer(max_epochs=1, accumulate_grad_batches=acc)
    trainer.fit(net, train_dataloaders=batches, max_steps=steps)
    got = trainer.train_losses
    assert len(got) == len(ref_losses) == acc * steps
    rel = [abs(a - b) / abs(b) for a, b in zip(got, ref_losses)]
    print("denseblock loss curve (device / oracle):", [f"{a:.5f}/{b:.5f}" for a, b in zip(got, ref_losses)])
    assert max(rel[:acc]) < 1e-5 and max(rel) < 1e-2, rel
    dev_sd = net.state_dict()
    diffs = torch.cat([(dev_sd[k].cpu().double() - v.detach()).abs().reshape(-1) for k, v in ref_sd.items()
                       if v.is_floating_point() and "running" not in k])
    assert float(diffs.max()) <= 2.05 * steps * 3e-4 and float(diffs.mean()) < 0.2 * steps * 3e-4
    for k, v in ref_sd.items():
        if "running" in k:
            assert float((dev_sd[k].cpu().double() - v).abs().max()) <= 2e-2 * max(1.0, float(v.abs().max())), k
    assert int(dev_sd["model.1.p2p.layer.0.denselayer1.norm1.num_batches_tracked"]) == 2 * acc * steps
    assert int(dev_sd["model.1.p2p.layer.0.denselayer1.norm2.num_batches_tracked"]) == acc * steps
    net.eval()
    with torch.no_grad():
        key, tonic, genre = net(batches[0]["mel"].to(DEV), batches[0]["seq_length"].to(DEV))
    ref = pcnet_oracle.pcnet_forward({k: v.detach() for k, v in ref_sd.items()}, batches[0]["mel"].double(), batches[0]["seq_length"])
    assert float((key.cpu().double() - ref[0]).abs().max()) < 2e-2 and torch.isfinite(tonic).all() and torch.isfinite(genre).all()


# ------------------------------------------------------------------------------------------------------ data parallel
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import ake_amd.distributed as D
    D.init_from_env("gloo")                                        # both ranks share cuda:0; RCCL needs one GPU per rank
    gold = load_golden("pcnet_default.npz")
    net, _ = default_net(gold)
    if rank == 1:                                                  # rank 1 starts from different weights: the broadcast must fix that
        with torch.no_grad():
            for p in net.parameters():
                p.add_(0.01)
    net = net.to(DEV)
    full = [make_batch(6, 40, 200 + i) for i in range(3)]
    lo, hi = D.shard_range(6, rank, world)
    mine = [{k: v[lo:hi] for k, v in b.items()} for b in full]
    trainer = ake_amd.Trainer(max_epochs=1, accumulate_grad_batches=1)
    trainer.fit(net, train_dataloaders=mine)
    flat, _ = net.flat_parameters()
    torch.save(flat.cpu(), os.path.join(out_dir, f"flat{rank}.pt"))


def test_two_rank_data_parallel_step(tmp_path, gold_default):
    mp.spawn(_dp_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    f0, f1 = torch.load(tmp_path / "flat0.pt"), torch.load(tmp_path / "flat1.pt")
    # running statistics are rank-local (BatchNorm is not synchronised); everything trainable must be identical
    net, _ = default_net(gold_default)
    net = net.to(DEV)
    flat, _ = net.flat_parameters()
    trainable = torch.zeros(flat.numel(), dtype=torch.bool)
    names = dict(net.named_parameters())
    for name, off, cnt in net._layout():
        if name in names:
            trainable[off:off + cnt] = True
    assert torch.equal(f0[trainable], f1[trainable])
    # single-process emulation: per step, the mean of the two shard gradients, then the same fused Adam
    full = [make_batch(6, 40, 200 + i) for i in range(3)]
    optim = net.configure_optimizers()[0][0]
    net.train()
    for b in full:
        optim.zero_grad()
        for lo, hi in ((0, 3), (3, 6)):
            shard = {k: v[lo:hi].to(DEV) for k, v in b.items()}
            net.training_step(shard, 0)["loss"].backward()
        optim.grad_scale = 0.5
        optim.step()
    flat, _ = net.flat_parameters()
    # every cross-workgroup reduction of the training path is a fixed-point (integer) sum, so a step does not depend on the order in
    # which workgroups arrive: the two-rank run (sum of the shard gradients by all-reduce, x 1/2 in the Adam kernel) and its
    # single-process emulation (the shard gradients accumulated into one buffer, x 1/2) agree BIT FOR BIT
    assert torch.equal(flat.cpu()[trainable], f0[trainable])


def test_training_step_is_deterministic(gold_default):
    """Two runs of the same three optimizer steps from the same weights give bit-identical gradients and weights (the
    float-atomic weight-gradient sums made training differ in the last digits from run to run)."""
    def run():
        net, _ = default_net(gold_default)
        net = net.to(DEV).train()
        optim = net.configure_optimizers()[0][0]
        grads = None
        for i in range(3):
            b = {k: v.to(DEV) for k, v in make_batch(6, 40, 300 + i).items()}
            optim.zero_grad()
            net.training_step(b, i)["loss"].backward()
            if i == 0:
                grads = net.flat_parameters()[1].clone()
            optim.step()
        return grads.cpu(), net.flat_parameters()[0].clone().cpu()
    g1, w1 = run()
    g2, w2 = run()
    assert torch.equal(g1, g2) and torch.equal(w1, w2)
    assert float(g1.abs().max()) > 0


def test_training_only_weight_load_and_the_eval_fragments(gold_default):
    """ake_pcnet_load_for_training_f32 (what a training step's weight sync calls: half of the repack launches) leaves the inference kernels'
    MFMA fragments stale: the C ABI refuses inference on such a handle (AKE_ERR_STATE), and the module's next eval-mode use reloads in full --
    outputs after three optimizer steps equal those of a fresh module loaded with the trained state_dict."""
    net, opt = default_net(gold_default)
    net = net.to(DEV).train()
    optim = net.configure_optimizers()[0][0]
    for i in range(3):
        b = {k: v.to(DEV) for k, v in make_batch(4, 40, 500 + i).items()}
        optim.zero_grad()
        net.training_step(b, i)["loss"].backward()
        optim.step()
    b = make_batch(4, 40, 600)
    net.train()
    net.training_step({k: v.to(DEV) for k, v in b.items()}, 0)      # syncs the handle with the training-only load
    assert net._h_eval_stale
    L = _lib.lib()
    x = b["mel"].to(DEV).contiguous()
    seq = b["seq_length"].to(DEV)
    outs = [torch.empty((4, 12), device=DEV), torch.empty((4, 12), device=DEV), torch.empty((4, 11), device=DEV)]
    ws = torch.empty(int(L.ake_pcnet_workspace_bytes(net.handle, 4, 40)), dtype=torch.uint8, device=DEV)
    rc = L.ake_pcnet_forward_f32(net.handle, x.data_ptr(), 4, 40, seq.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(),
                                 ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == -3 and b"ake_pcnet_load_from_device_f32" in L.ake_last_error()
    net.eval()
    with torch.no_grad():
        got = net(x, seq)
    assert not net._h_eval_stale
    fresh, _ = default_net(gold_default)
    fresh.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()}, strict=True)
    fresh = fresh.to(DEV).eval()
    with torch.no_grad():
        ref = fresh(x, seq)
    for a, r in zip(got, ref):
        assert torch.equal(a, r)


def test_mixed_precision_inference_on_trained_weights():
    """the f16 single-product convolutions were only ever checked on seeded-random weights.  Train the default net on a
    GiantSteps-shaped synthetic set (config 3 in small: 160 clips of 15 s, batch 8, accumulate 2, 4 epochs -- the filters have structure, the
    BatchNorm statistics are real), then eval-mode inference of the TRAINED state_dict on 64 clips in both precisions against the float64
    oracle on the same log-CQT.

    Measured (round 3): 'f32x3' 1e-6 .. 6e-6; 'mixed' 4e-4 .. 8e-4 -- INSIDE the 1e-3 budget but 30x the 2e-5 it shows on seeded-random
    weights: trained filters difference nearly equal inputs, so the unbiased 2^-12 operand roundings of the f16 single-product kernels no
    longer average out against the output.  Diagnostic build, same trained net: pitch convolutions on exact f32 (AKE_P2P_F32) 6.7e-4, layer 0 +
    pitch-class kernels on exact f32 (AKE_PC_F32) 4.2e-4, both 1e-6: layer 0's f16 stack contributes ~6.7e-4, the pitch stack ~4.2e-4.
    So: 'mixed' is asserted against the BUDGET (1e-3) here, and who needs margin on a trained net asks for opt.precision = 'f32x3'."""
    opt = Namespace(conv_layers=3, n_filters=4, head_layers=2, time_pool_size=2, genre=True, max_pool=False, frames=5, octaves=8, lr=1e-3,
                    gamma=0.96, acc_grad=2, reg=0, key_weight=1.0, tonic_weight=1.0, genre_weight=0.1, use_cos=False, no_ckpt=True, local=False,
                    only_semitones=False, multi_scale=False)
    import random
    random.seed(5)                                                    # (import_data shuffles with `random`, the DataLoader with torch's generator)
    train = ake_amd.KeyDataset(True, opt)
    train.import_data(ake_amd.SyntheticSineMixLoader(160), shuffle=True)
    torch.manual_seed(3)
    net = ake_amd.PitchClassNet(288, 12, 2, 7, opt, batch_size=8, train_set=train, val_set=None).to(DEV)
    sd0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    trainer = ake_amd.Trainer(max_epochs=4, accumulate_grad_batches=2)
    trainer.fit(net)
    losses = trainer.train_losses
    assert np.mean(losses[-10:]) < np.mean(losses[:10]) - 0.3, (np.mean(losses[:10]), np.mean(losses[-10:]))
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    moved = max(float((sd[k] - sd0[k]).abs().max()) for k in sd if k.endswith(".weight"))
    assert moved > 1e-2                                               # the weights are not the initial ones any more
    items = [train[i] for i in range(64)]
    mel = torch.stack([torch.as_tensor(it["mel"]) for it in items]).to(DEV).float()
    seq = torch.stack([torch.as_tensor(it["seq_length"]) for it in items]).to(DEV)
    ref = pcnet_oracle.pcnet_forward(pcnet_oracle.to_dtype(sd, torch.float64), mel.double().cpu(), seq.cpu())
    errs = {}
    for prec in ("mixed", "f32x3"):
        o = Namespace(**vars(opt), precision=prec)
        m = ake_amd.PitchClassNet(288, 12, 2, 7, o)
        m.load_state_dict(sd, strict=True)
        with torch.no_grad():
            out = m.to(DEV).eval()(mel, seq)
        errs[prec] = [float((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-6)) for a, b in zip(out, ref)]
    print("\ntrained weights, rel err (key, tonic, genre) vs float64:", errs)
    assert max(errs["mixed"]) < 1e-3 and max(errs["f32x3"]) < 2e-5, errs
