"""Synthetic VoxCeleb2-shaped samples with the output contract of the reference's
dataloaders/voxceleb2_segmentation_nolandmarks.py:182-248: all fp32 in [0,1], NCHW.
  data_dict:   enc_rgbs K x 3 x S x S (K=8 meta-train, 1 fine-tune), pose_input_rgbs 1 x 3 x S x S, target_rgbs 1 x 3 x S x S
  target_dict: real_segm 1 x 3 x S x S (one mask expanded to 3 channels), label int64 (0 when fine-tuning)
There is no network / dataset in the build image; real VoxCeleb2 loading is out of scope."""
import math

import torch


def blob_mask(size, cx, cy, radius, soft):
    ys, xs = torch.meshgrid(torch.arange(size, dtype=torch.float32), torch.arange(size, dtype=torch.float32), indexing='ij')
    d = torch.sqrt((xs - cx) ** 2 + (ys - cy) ** 2)
    return torch.sigmoid((radius - d) / soft)


def make_sample(index, image_size, num_frames, num_labels, finetune, seed):
    g = torch.Generator().manual_seed(seed * 1000003 + index)
    s = image_size
    mask = blob_mask(s, s * (0.4 + 0.2 * torch.rand(1, generator=g).item()), s * (0.4 + 0.2 * torch.rand(1, generator=g).item()),
                     s * 0.3, s * 0.02)
    image = torch.rand(3, s, s, generator=g)
    data = {'enc_rgbs': torch.rand(num_frames, 3, s, s, generator=g),
            'pose_input_rgbs': torch.rand(1, 3, s, s, generator=g),
            'target_rgbs': (image * mask)[None]}
    label = 0 if finetune else int(torch.randint(0, num_labels, (1,), generator=g))
    target = {'real_segm': mask[None, None].expand(1, 3, s, s).contiguous(), 'label': label}
    return data, target


class Dataset(torch.utils.data.Dataset):
    @staticmethod
    def get_args(parser):
        parser.add('--num_labels', type=int, default=98000, help='identities in the training split (data/splits/train.csv)')
        parser.add('--synthetic_dataset_len', type=int, default=64)
        parser.add('--n_frames_for_encoder', type=int, default=8)
        return parser

    @staticmethod
    def get_dataset(args, part):
        return Dataset(args, part)

    def __init__(self, args, part):
        self.size = args.image_size
        self.finetune = bool(getattr(args, 'finetune', False))
        self.frames = 1 if self.finetune else args.n_frames_for_encoder
        self.num_labels = args.num_labels
        self.length = int(math.ceil(args.synthetic_dataset_len / args.world_size) * args.world_size)
        self.seed = (args.random_seed or 0) + (0 if part == 'train' else 7919)

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        return make_sample(index, self.size, self.frames, self.num_labels, self.finetune, self.seed)
