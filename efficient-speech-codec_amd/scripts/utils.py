"""Data plumbing of the evaluation harness (reference: scripts/utils.py:27-46,75-91), wav I/O through scipy."""
import glob

import numpy as np
import torch
import yaml
from scipy.io import wavfile
from torch.utils.data import Dataset


def load_wav(path):
    sr, pcm = wavfile.read(path)
    if pcm.dtype == np.int16:
        x = pcm.astype(np.float32) / 32768.0
    elif pcm.dtype == np.int32:
        x = pcm.astype(np.float32) / 2147483648.0
    else:
        x = pcm.astype(np.float32)
    x = torch.from_numpy(np.atleast_2d(x.T if x.ndim == 2 else x))          # (channels, L) like torchaudio.load
    return x, sr


class EvalSet(Dataset):
    """utils.py:27-40 -- first channel, last 80 samples dropped (so that a 3 s file gives an even frame count)."""

    def __init__(self, eval_folder_path):
        self.testset_files = sorted(glob.glob(f"{eval_folder_path}/*.wav")) or sorted(glob.glob(f"{eval_folder_path}/*/*.wav"))
        self.testset_files = self.testset_files[:180000]

    def __len__(self):
        return len(self.testset_files)

    def __getitem__(self, i):
        x, _ = load_wav(self.testset_files[i])
        return x[0, :-80]


def read_yaml(pth):
    with open(pth, "r") as f:
        return yaml.safe_load(f)
