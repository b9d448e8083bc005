"""``data_provider(args, flag) -> (Dataset, DataLoader)`` -- the loader contract of
IGN/data_factory/data_factory.py:29-121 for the classification task.

Batches are ``(X[B, T, C] float32, y[B, 1], mask[B, T] bool)``.  Registry keys follow the reference
(``UEA``, ``EEG``, ``EEG3``); ``SYNTH`` is the synthetic benchmark provider.  Under ``torch.distributed`` each
rank iterates a disjoint, equally sized slice of a shared permutation (ign_hip.ddp.shard_indices).
"""
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, Sampler

from data_provider.data_loader import UEAloader
from data_provider.eeg_npy import EEGNpyDataset, EEGNpyDataset3Class, collate_raw
from data_provider.synthetic import SyntheticEEG
from data_provider.uea import collate_fn

data_dict = {
    'UEA': UEAloader,
    'EEG': EEGNpyDataset,           # 39 classes
    'EEG3': EEGNpyDataset3Class,    # 3 classes
    'SYNTH': SyntheticEEG,
}


class RankShardSampler(Sampler):
    """Global permutation from a shared seed; rank r takes the r-th contiguous slice."""

    def __init__(self, n, rank, world, shuffle, seed=0):
        self.n, self.rank, self.world, self.shuffle, self.seed, self.epoch = n, rank, world, shuffle, seed, 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __iter__(self):
        from ign_hip.ddp import shard_indices
        self.epoch += 1
        return iter(shard_indices(self.n, self.rank, self.world, self.epoch, self.seed, self.shuffle).tolist())

    def __len__(self):
        return self.n // self.world


def data_provider(args, flag, bin_edges=None):
    flag = flag.lower()
    if args.task_name != 'classification':
        raise NotImplementedError(f"task_name={args.task_name!r}: only the classification path is rebuilt "
                                  f"(SURVEY section 2 marks forecasting / anomaly / regression out of scope)")
    if args.data not in data_dict:
        raise KeyError(f"--data {args.data!r} not in {sorted(data_dict)}")
    Data = data_dict[args.data]
    shuffle = flag != 'test'
    if args.data == 'SYNTH':
        shape = getattr(args, 'synthetic', None) or "8192,122,1000,3"
        n, C, T, N = (int(v) for v in shape.split(','))
        data_set = Data(flag=flag, n=n if flag == 'train' else max(args.batch_size, n // 8), seq_len=T, enc_in=C,
                        num_classes=N)
    elif args.data in ('EEG', 'EEG3'):
        # on a GPU the items stay RAW and the batch is standardised + transposed on the device (device_prefetch.py)
        raw = bool(getattr(args, 'device_standardise', torch.cuda.is_available()))
        data_set = Data(root_path=args.root_path, flag=flag, test_size=getattr(args, 'test_size', 0.2),
                        val_size=getattr(args, 'val_size', 0.1), raw=raw)
    else:
        data_set = Data(root_path=args.root_path, flag=flag)

    max_len = getattr(data_set, 'seq_len', None) or getattr(args, 'seq_len', None)
    sampler = None
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and flag == 'train':
        sampler = RankShardSampler(len(data_set), dist.get_rank(), dist.get_world_size(), shuffle,
                                   seed=max(0, getattr(args, 'seed', 0)))
    raw = bool(getattr(data_set, 'raw', False))
    if args.data == 'SYNTH' and args.num_workers == 0:
        # samples are rows of one in-memory tensor: a batch is one multi-threaded gather into a pinned buffer
        from data_provider.device_prefetch import TensorBatchLoader
        return data_set, TensorBatchLoader(data_set, args.batch_size, shuffle=shuffle, sampler=sampler,
                                           pin_memory=torch.cuda.is_available())
    loader = DataLoader(data_set, batch_size=args.batch_size, shuffle=(shuffle and sampler is None), sampler=sampler,
                        num_workers=args.num_workers, drop_last=False, pin_memory=torch.cuda.is_available(),
                        collate_fn=collate_raw if raw else (lambda b: collate_fn(b, max_len=max_len)))
    loader.device_transform = 'standardise_raw' if raw else None
    return data_set, loader
