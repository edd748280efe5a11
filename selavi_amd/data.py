"""Synthetic stand-in for the reference's AVideoDataset output contract
(datasets/AVideoDataset.py:355-454): ``dataset[i] -> (frames[3,T,H,W], spec[1,F,T'], label, index, vid_idx)``.
Deterministic per index (counter-based), generated on the host; used by tests, smoke and bench."""
import torch


class SyntheticAVDataset(torch.utils.data.Dataset):
    def __init__(self, n=3328, T=8, S=112, F=40, Tp=100, n_classes=28, seed=31, device=None):
        self.n, self.T, self.S, self.F, self.Tp, self.seed = n, T, S, F, Tp, seed
        g = torch.Generator().manual_seed(seed)
        self._labels = torch.randint(0, n_classes, (n,), generator=g).tolist()
        self.valid_indices = list(range(n))
        self.n_classes = n_classes
        self.device = device

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + int(i))
        lab = self._labels[i]
        # class-dependent mean so that clusters are learnable, unit variance like the normalised real data
        video = torch.randn(3, self.T, self.S, self.S, generator=g) + 0.25 * ((lab % 7) - 3)
        audio = torch.randn(1, self.F, self.Tp, generator=g) + 0.25 * ((lab % 5) - 2)
        return video, audio, lab, i, i
