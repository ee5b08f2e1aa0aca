"""Dataloader front-end with the reference's surface (dataloaders/dataloader.py:10-50): ``Dataloader(name)`` resolves
``dataloaders.<name>.Dataset``, adds ``--num_workers/--prefetch_size/--batch_size`` and builds a loader whose per-GPU
batch is ``batch_size // num_gpus`` over a rank-strided subset.  The reference's prefetching subclass pokes torch-1.5
private iterator classes and its VoxCeleb2 datasets need cv2/imgaug/pandas: host-side I/O, out of scope here
 -- only the output contract matters, which ``synthetic_voxceleb2`` reproduces."""
import logging

import torch
from torch.utils.data import DataLoader

from latent_pose_reenactment_amd.utils.utils import load_module

logger = logging.getLogger('dataloaders.dataloader')


class Dataloader:
    def __init__(self, dataset_name):
        self.dataset = load_module('dataloaders', dataset_name).Dataset

    def get_args(self, parser):
        parser.add('--num_workers', type=int, default=4, help='Number of data loading workers.')
        parser.add('--prefetch_size', type=int, default=16, help='Prefetch queue size')
        parser.add('--batch_size', type=int, default=64, help='Batch size')
        return self.dataset.get_args(parser)

    def get_dataloader(self, args, part, phase):
        if hasattr(self.dataset, 'get_dataloader'):
            return self.dataset.get_dataloader(args, part)
        dataset = self.dataset.get_dataset(args, part)
        assert len(dataset) % args.world_size == 0, \
            "`dataset.get_dataset()` was expected to return a dataset equally divisible by `args.world_size`"
        dataset = torch.utils.data.Subset(dataset, range(args.rank, len(dataset), args.world_size))
        logger.info(f"This process will receive a dataset with {len(dataset)} samples")
        if len(dataset) < args.batch_size:
            logger.warning(f"Dataset length is smaller than batch size ({len(dataset)} < {args.batch_size}), reducing the latter")
            args.batch_size = len(dataset)
        return DataLoader(dataset, batch_size=args.batch_size // args.num_gpus, num_workers=0, pin_memory=True,
                          drop_last=(phase == 'train'), shuffle=(part == 'train'))
