"""Drop-in ``KeyDataset`` whose CQT stage runs on the GPU.

Keeps the reference's dataset surface (KeyDataset.py:32-264): ``KeyDataset(genre, opt)``,
``import_data(*loaders)``, ``len()``, ``ds[i]`` -> the item dict of KeyDataset.py:242-256 with the
same keys, dtypes and zero padding to ``seq_length_max``; loaders expose ``name``,
``get_filenames()`` and ``get_all(file, pitch_shift, genre, opt, multi_scale)``.

What differs, on purpose: the reference computes one CQT per file on a CPU thread inside
``get_all`` (KeyDataset.py:488-495) and caches it on disk.  Here loaders hand over *waveforms*
and ``import_data`` batches clips of equal length through ``ake_cqt_logmag_f32``; nothing is
cached on disk.  File-system loaders for the 14 public datasets, annotation parsing and audio
decoding (KeyDataset.py:268-466, 514-1233) are out of scope: ``SyntheticSineMixLoader`` and
``WaveformLoader`` supply clips instead.
"""
from __future__ import annotations

import random
from collections import defaultdict

import numpy as np
import torch

from . import synthetic
from .cqt import get_plan, hop_for

SIGNATURE = [n + " minor" for n in ("C", "Db", "D", "Eb", "E", "F", "Gb", "G", "Ab", "A", "Bb", "B")] + \
            [n + " major" for n in ("C", "Db", "D", "Eb", "E", "F", "Gb", "G", "Ab", "A", "Bb", "B")]   # KeyDataset.py:524-527
_ENHARMONIC = {"C#": "Db", "D#": "Eb", "F#": "Gb", "G#": "Ab", "A#": "Bb", "Cb": "B", "E#": "F", "B#": "C", "Fb": "E"}


def signature_id(key_name: str) -> int:
    """'F# minor' / 'Bb major' -> index into the 24-way ``signature`` list (0..11 minor, 12..23 major)."""
    tonic, mode = key_name.strip().split()
    tonic = _ENHARMONIC.get(tonic, tonic)
    return SIGNATURE.index(f"{tonic} {mode}")


def labels_for_signature(sig: int, genre_id: int | None, genre_flag: bool):
    """Label tensors of KeyDataset.py:443-454 for key ``sig``; genre is zeros(8) when the flag is off (:476)."""
    key_labels = torch.from_numpy(synthetic.key_pitch_classes(sig))
    key_signature_id = torch.nn.functional.one_hot(torch.tensor(sig), 24).float()
    tonic = torch.nn.functional.one_hot(torch.tensor(sig % 12), 12).float()
    if genre_flag:
        genre = torch.zeros(synthetic.N_GENRES)
        if genre_id is not None and genre_id >= 0:
            genre[genre_id] = 1.0
    else:
        genre = torch.zeros(8)
    return key_labels, key_signature_id, genre, tonic


class DatasetLoader:
    """Loader protocol (KeyDataset.py:268-313): subclasses supply clips and key names."""

    def __init__(self, dataset_loc=None):
        self.name = None
        self.dataset_loc = dataset_loc
        self.size = -1

    def get_filenames(self):
        raise NotImplementedError("The standard Dataset Loader has no allocated Dataset")

    def get_size(self):
        return self.size

    def get_waveform(self, file_id):
        """-> (waveform float32 (n,), sample_rate)"""
        raise NotImplementedError

    def get_key_signature_id(self, file_id) -> int:
        raise NotImplementedError

    def get_genre_id(self, file_id):
        return None

    def get_all(self, file_id, pitch_shift, genre, opt, multi_scale=False):
        """Single-clip path with the reference's return order (KeyDataset.py:469-509)."""
        wav, sr = self.get_waveform(file_id)
        mel = cqt_features(torch.as_tensor(wav)[None], sr, opt)[0]
        key_labels, key_signature_id, g, tonic = labels_for_signature(self.get_key_signature_id(file_id), self.get_genre_id(file_id), genre)
        return mel.reshape(1, mel.shape[0], mel.shape[1]).double().cpu(), key_labels, key_signature_id, g, tonic


class SyntheticSineMixLoader(DatasetLoader):
    """Seeded sine-mix clips (synthetic.py): clip i has key ``i % 24`` and genre ``i % 11``."""

    def __init__(self, n_clips, first=0, n_samples=synthetic.N_SAMPLES, sample_rate=synthetic.SR, name="Synthetic SineMix"):
        super().__init__(None)
        self.name = name
        self.size = n_clips
        self.first, self.n_samples, self.sample_rate = first, n_samples, sample_rate

    def get_filenames(self):
        return list(range(self.first, self.first + self.size))

    def get_waveform(self, file_id):
        return synthetic.make_clip(int(file_id), self.n_samples, self.sample_rate)[0], self.sample_rate

    def get_key_signature_id(self, file_id):
        return int(file_id) % 24

    def get_genre_id(self, file_id):
        return int(file_id) % synthetic.N_GENRES


class WaveformLoader(DatasetLoader):
    """In-memory clips: ``waveforms`` list of 1-D arrays, ``keys`` like 'A minor', optional genre ids."""

    def __init__(self, name, waveforms, keys, sample_rate, genres=None):
        super().__init__(None)
        self.name, self.sample_rate = name, sample_rate
        self.waveforms, self.keys, self.genres = waveforms, keys, genres
        self.size = len(waveforms)

    def get_filenames(self):
        return list(range(self.size))

    def get_waveform(self, file_id):
        return np.asarray(self.waveforms[file_id], dtype=np.float32), self.sample_rate

    def get_key_signature_id(self, file_id):
        k = self.keys[file_id]
        return k if isinstance(k, int) else signature_id(k)

    def get_genre_id(self, file_id):
        return None if self.genres is None else self.genres[file_id]


def cqt_features(waveforms: torch.Tensor, rate: int, opt, lengths=None) -> torch.Tensor:
    """(B, n) waveforms -> log-CQT (B, 36*octaves, T) float32 on the GPU (KeyDataset.py:485-499); ``lengths``: samples per row of
    a ragged batch (clip i then has ``1 + lengths[i] // hop`` frames, zeros after them)."""
    frames = getattr(opt, "frames", 5)
    if frames <= 0:
        raise NotImplementedError("opt.frames == 0 (fixed 592-frame windows, KeyDataset.py:490,501-503) is not built")
    if getattr(opt, "only_semitones", False) or getattr(opt, "multi_scale", False):
        raise NotImplementedError("--only_semitones / --multi_scale CQTs are not built (SURVEY.md section 2.1)")
    # opt.cqt_q_mode (not a reference option): 1 selects librosa <= 0.9's filter Q, see ake_amd.cqt.get_plan
    plan = get_plan(rate, hop_for(rate, frames), 36 * getattr(opt, "octaves", 8), 36, q_mode=int(getattr(opt, "cqt_q_mode", 0)))
    return plan.logmag(waveforms, lengths=lengths)


class KeyDataset:

    def __init__(self, genre, opt, cqt_batch=64):
        self.datasets = {}
        self.filenames = []
        self.genre = genre
        self.mel, self.mel2 = {}, {}
        self.key_labels, self.key_signature_id, self.genre_labels, self.tonic_labels = {}, {}, {}, {}
        self.opt = opt
        self.seq_length_max = 0
        self.cqt_batch = cqt_batch
        if getattr(opt, "local", False):
            raise NotImplementedError("--local datasets (per-frame labels, KeyDataset.py:443-454) are not built; PitchClassNet(opt.local) "
                                      "inference is (SURVEY.md section 8f)")

    def __len__(self):
        return len(self.filenames)

    def load_files(self, *dataset_loaders):
        self.filenames = []
        for loader in dataset_loaders:
            if not isinstance(loader, DatasetLoader):
                continue
            for f in loader.get_filenames():
                self.filenames.append((f, loader.name, torch.tensor(0)))             # KeyDataset.py:64-76

    def load_dataset_handler(self, *dataset_loaders):
        for loader in dataset_loaders:
            if isinstance(loader, DatasetLoader):
                self.datasets[loader.name] = loader

    def import_data(self, *dataset_loaders, shuffle=True):
        self.load_files(*dataset_loaders)
        self.load_dataset_handler(*dataset_loaders)
        if shuffle:
            random.shuffle(self.filenames)                                            # KeyDataset.py:97
        self.store_content()
        self.find_longest_seq()
        print("Length of Data: " + str(len(self.mel)))

    def store_content(self):
        """All clips -> CQT on the GPU, batched by sample rate: clips of different lengths share a launch (ragged batch, sorted by
        length so that a batch pads little); every clip keeps its own frame count; labels per clip (KeyDataset.py:121-138)."""
        groups = defaultdict(list)
        waves = {}
        for idx, (f, dname, _) in enumerate(self.filenames):
            wav, sr = self.datasets[dname].get_waveform(f)
            waves[idx] = torch.as_tensor(wav, dtype=torch.float32).reshape(-1)
            groups[sr].append(idx)
        for sr, idxs in groups.items():
            idxs = sorted(idxs, key=lambda i: waves[i].numel())
            hop = hop_for(sr, getattr(self.opt, "frames", 5))
            for s in range(0, len(idxs), self.cqt_batch):
                part = idxs[s:s + self.cqt_batch]
                lens = [waves[i].numel() for i in part]
                if min(lens) == max(lens):
                    mel = cqt_features(torch.stack([waves[i] for i in part]), sr, self.opt)
                else:
                    batch = torch.zeros((len(part), max(lens)), dtype=torch.float32)
                    for j, i in enumerate(part):
                        batch[j, :lens[j]] = waves[i]
                    mel = cqt_features(batch, sr, self.opt, lengths=torch.tensor(lens, dtype=torch.int64))
                mel = mel.double().cpu()                                               # .double(): KeyDataset.py:509
                for j, i in enumerate(part):
                    self.mel[str(i)] = mel[j:j + 1, :, :1 + lens[j] // hop].clone()    # (1, bins, T_i)
                    self.mel2[str(i)] = None
        for idx, (f, dname, _) in enumerate(self.filenames):
            ld = self.datasets[dname]
            kl, ks, g, t = labels_for_signature(ld.get_key_signature_id(f), ld.get_genre_id(f), self.genre)
            self.key_labels[str(idx)], self.key_signature_id[str(idx)] = kl, ks
            self.genre_labels[str(idx)], self.tonic_labels[str(idx)] = g, t
        print("done", flush=True)

    def find_longest_seq(self):
        for i in range(len(self.mel)):                                                # KeyDataset.py:113-119
            self.seq_length_max = max(self.seq_length_max, self.mel[str(i)].shape[2])
        print("Max. Seq. Length: " + str(self.seq_length_max))

    def __getitem__(self, idx):
        mel = self.mel[str(idx)]
        seq_length = mel.shape[2]
        pad = self.seq_length_max - seq_length
        padded = torch.cat((mel, torch.zeros([mel.shape[0], mel.shape[1], pad], dtype=mel.dtype)), dim=2)   # KeyDataset.py:243-245
        return {"mel": padded, "key_labels": self.key_labels[str(idx)], "tonic_labels": self.tonic_labels[str(idx)],
                "key_signature_id": self.key_signature_id[str(idx)], "genre": self.genre_labels[str(idx)],
                "seq_length": seq_length}                                              # KeyDataset.py:249-256
