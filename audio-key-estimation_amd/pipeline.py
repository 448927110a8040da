"""Whole hot path in one call: waveforms on the GPU -> (key, tonic[, genre]).

Host wrapper of ``ake_pipeline_forward_f32``: CQT, seq_length fill and the network run back
to back on one stream with no host round trip (what ``DatasetLoader.get_all`` ->
``KeyDataset.__getitem__`` -> ``PitchClassNet.forward`` do in the reference, KeyDataset.py:469-509,
242-256, models.py:846).
"""
from __future__ import annotations

import torch

from . import _lib
from .audio import get_resampler
from .cqt import CQTPlan, hop_for
from .models import PitchClassNet


class KeyEstimator:
    """``streams`` > 1: consecutive calls are issued round-robin on that many side streams, each with a workspace of its own, so
    that independent batches overlap on the GPU -- the CQT stage is VALU / HBM-bound, the network MFMA-bound, and one batch's CQT
    runs under another's convolutions (measured: +8..12 % clips/s at 2 streams).  The weights and CQT tables are shared (read-only
    during a forward).  Outputs of such calls belong to their side stream: call ``join()`` before the caller's stream reads them.

    ``wrap_mode``: the pitch convolutions are circular in TIME as well (models.py:221,230), and a batch is zero-padded to its
    longest clip (the reference pads to the longest clip of the whole dataset, KeyDataset.py:243-245), so a shorter clip's last
    frames wrap into padding and its outputs depend on what it was batched with.  ``"dataset_max"`` (default) is that
    behaviour, bit for bit.  ``"true_end"`` (opt-in) wraps every clip at its OWN last frame: clips are
    grouped by frame count and each group runs unpadded, so a clip's outputs are those of the clip alone."""

    def __init__(self, net: PitchClassNet, sample_rate: int = 22050, frames: int = 5, streams: int = 1, wrap_mode: str = "dataset_max",
                 q_mode: int = 0):
        if wrap_mode not in ("dataset_max", "true_end"):
            raise ValueError("wrap_mode must be 'dataset_max' or 'true_end'")
        self.net = net.eval()
        self.device = net._device()
        self.sample_rate, self.wrap_mode = int(sample_rate), wrap_mode
        self.plan = CQTPlan(sample_rate, hop_for(sample_rate, frames), net.pitches, 36, q_mode=q_mode, device=self.device)   # q_mode: ake_amd.cqt.get_plan
        self.streams = max(1, int(streams))
        self._slots = [{"ws": None, "stream": None} for _ in range(self.streams)]
        self._turn = 0

    def join(self):
        """Make the caller's current stream wait for every call issued so far (``streams`` > 1; a no-op otherwise)."""
        if self.streams == 1:
            return
        cur = torch.cuda.current_stream(self.device)
        for slot in self._slots:
            if slot["stream"] is not None:
                cur.wait_stream(slot["stream"])

    @torch.no_grad()
    def __call__(self, audio: torch.Tensor, lengths: torch.Tensor | None = None, rate: int | None = None, channel: int = 0):
        """audio (B, n) or (B, C, n) float32 on the GPU -> tuple of (B,12), (B,12)[, (B,11)] float32 tensors.

        ``lengths`` (B,) int64: ragged batch, row i holds ``lengths[i] <= n`` samples; every clip is pooled over its own frames
        (``seq_length`` = ``1 + lengths[i] // hop``), as a ``KeyDataset`` batch of unequal clips is (KeyDataset.py:245-256).
        ``rate``: sample rate of ``audio`` when it is not the estimator's -- it is resampled on the device first
        (``scipy.signal.resample_poly``'s filter); ``channel``: which channel of (B, C, n) audio to take (0 = the reference's
        ``waveform[0]``, KeyDataset.py:480) or -1 for the mean of all."""
        self.net._sync_weights(self.device, for_eval=True)
        if audio.dim() == 3 or (rate is not None and int(rate) != self.sample_rate):
            rs = get_resampler(self.sample_rate if rate is None else int(rate), self.sample_rate, self.device)
            audio, len_out = rs(audio, channel=channel, lengths=lengths)
            lengths = len_out if lengths is not None else None
        slot = self._slots[self._turn]
        if self.streams == 1:
            return self._run_wrapped(slot, audio, lengths)
        self._turn = (self._turn + 1) % self.streams
        with torch.cuda.device(self.device):
            if slot["stream"] is None:
                slot["stream"] = torch.cuda.Stream(self.device)
            slot["stream"].wait_stream(torch.cuda.current_stream(self.device))      # the inputs were produced on the caller's stream
            cur = torch.cuda.current_stream(self.device)
            with torch.cuda.stream(slot["stream"]):
                out = self._run_wrapped(slot, audio, lengths)
            # the outputs were allocated on the side stream and will be read on the caller's: tell the allocator now (nothing to
            # remember until join(), nothing to drop when a caller never joins)
            for t in out:
                if t is not None:
                    t.record_stream(cur)
            audio.record_stream(slot["stream"])
        return out

    def _run_wrapped(self, slot, audio, lengths):
        """wrap_mode "true_end": one unpadded call per distinct frame count (the clips of a group share T, so no frame is padding)."""
        if self.wrap_mode != "true_end" or lengths is None:
            return self._run(slot, audio, lengths)
        lengths = torch.as_tensor(lengths).to(device=self.device, dtype=torch.int64)
        lens = lengths.cpu()                                      # (opt-in mode: grouping needs the lengths on the host)
        frames = 1 + lens // self.plan.hop_length
        B = audio.shape[0]
        outs = [torch.empty((B, w), dtype=torch.float32, device=self.device) for w in ((12, 12, 11) if self.net.genre else (12, 12))]
        for t in torch.unique(frames).tolist():
            idx = torch.nonzero(frames == t).flatten()
            n_g = int(lens[idx].max())
            # a group's clips have the same frame count but not the same sample count: lengths stay (the CQT zero-pads the tails)
            sub = audio.index_select(0, idx.to(self.device))[:, :max(n_g, 1)].contiguous()
            if self.plan.num_frames(sub.shape[1]) != t:            # n_g rounds into the next hop only if a clip does: cannot happen
                raise _lib.AkeError("true_end grouping: frame count mismatch")
            got = self._run(slot, sub, lengths[idx.to(self.device)])
            for o, g in zip(outs, got):
                o.index_copy_(0, idx.to(self.device), g)
        return tuple(outs)

    def _run(self, slot, audio, lengths):
        net, L = self.net, _lib.lib()
        audio = audio.to(device=self.device, dtype=torch.float32)
        if audio.stride(-1) != 1:
            audio = audio.contiguous()
        B, n = audio.shape
        nbytes = L.ake_pipeline_workspace_bytes(self.plan.handle, net.handle, B, n)
        if slot["ws"] is None or slot["ws"].numel() < nbytes:
            slot["ws"] = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self.device)
        ws = slot["ws"]
        key = torch.empty((B, 12), dtype=torch.float32, device=self.device)
        tonic = torch.empty((B, 12), dtype=torch.float32, device=self.device)
        genre = torch.empty((B, 11), dtype=torch.float32, device=self.device) if net.genre else None
        with torch.cuda.device(self.device):
            if lengths is None:
                _lib.check(L.ake_pipeline_forward_f32(self.plan.handle, net.handle, audio.data_ptr(), B, n, audio.stride(0),
                                                      key.data_ptr(), tonic.data_ptr(), genre.data_ptr() if genre is not None else None,
                                                      ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream),
                           "ake_pipeline_forward_f32")
            else:
                lengths = torch.as_tensor(lengths).to(device=self.device, dtype=torch.int64).contiguous()
                assert lengths.shape == (B,)
                _lib.check(L.ake_pipeline_forward_ragged_f32(self.plan.handle, net.handle, audio.data_ptr(), B, n, audio.stride(0), lengths.data_ptr(),
                                                             key.data_ptr(), tonic.data_ptr(), genre.data_ptr() if genre is not None else None,
                                                             ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream),
                           "ake_pipeline_forward_ragged_f32")
        return (key, tonic, genre) if net.genre else (key, tonic)
