#!/usr/bin/env python3
"""Reenactment ("driving") entry point with the reference's CLI (drive.py:19-98): load a fine-tuned checkpoint, take the
EMA weights, and for every driving frame run ``embedder.get_pose_embedding`` + ``generator`` (HIP kernels, eval mode: the
spectral norms are constant so the packed bf16 weights are reused).  The reference writes an .mp4 through cv2 (absent here,
out of scope): frames are written as one .npy uint8 array [T, H, 2W, 3] (driver | result) per driving sequence instead."""
import argparse
import copy
import logging
import os
import sys
from pathlib import Path

HERE = os.path.dirname(os.path.abspath(__file__))
for _p in (HERE, os.path.dirname(HERE)):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_grad_enabled(False)

from utils import utils  # noqa: E402


def string_to_valid_filename(x):
    return str(x).replace('/', '_')


def load_for_inference(checkpoint_path, data_root, device):
    checkpoint_object = utils.torch_load(checkpoint_path)
    saved_args = copy.copy(checkpoint_object['args'])
    saved_args.finetune = True
    saved_args.inference = True
    saved_args.data_root = data_root
    saved_args.world_size = 1
    saved_args.num_workers = 1
    saved_args.batch_size = 1
    saved_args.device = device
    saved_args.bboxes_dir = Path("/non/existent/file")
    saved_args.prefetch_size = 4
    embedder, generator, _, running_averages, _, _, _ = utils.load_model_from_checkpoint(checkpoint_object, saved_args)
    if 'embedder' in running_averages:
        embedder.load_state_dict(running_averages['embedder'])
    if 'generator' in running_averages:
        generator.load_state_dict(running_averages['generator'])
    embedder.train(not saved_args.set_eval_mode_in_test)
    generator.train(not saved_args.set_eval_mode_in_test)
    return embedder, generator, saved_args


def drive_frames(embedder, generator, data_dict):
    """one iteration of the hot loop (drive.py:84-88) on a batch of B >= 1 driving frames -> uint8 [B, H, 2W, 3] frame grids
    (driver | result), left ON THE DEVICE: the caller gathers a whole sequence and crosses PCIe once (no per-frame host sync)"""
    embedder.get_pose_embedding(data_dict)
    generator(data_dict)
    to_u8 = lambda img: img.permute(0, 2, 3, 1).clamp(0, 1).mul(255).byte()
    return torch.cat((to_u8(data_dict['pose_input_rgbs'][:, 0]), to_u8(data_dict['fake_rgbs'])), dim=2)


def drive_frame(embedder, generator, data_dict):
    """single-frame form (B = 1) -> uint8 HWC frame grid"""
    return drive_frames(embedder, generator, data_dict)[0]


def main():
    logging.basicConfig(level=logging.INFO, stream=sys.stdout, format="%(asctime)s - %(levelname)s - %(message)s")
    logger = logging.getLogger('drive')
    ap = argparse.ArgumentParser(description="Render 'puppeteering' frames, given a fine-tuned model and driving images.",
                                 formatter_class=argparse.RawTextHelpFormatter)
    ap.add_argument('checkpoint_path', type=Path)
    ap.add_argument('data_root', type=Path)
    ap.add_argument('--images_paths', type=Path, nargs='+')
    ap.add_argument('--destination', type=Path, required=True)
    ap.add_argument('--batch_size', type=int, default=1, help='driving frames per generator call (the reference drives one frame at a time)')
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit('drive.py needs the MI355X: the generator has no CPU fallback')
    device = 'cuda:0'
    embedder, generator, saved_args = load_for_inference(args.checkpoint_path, args.data_root, device)
    saved_args.batch_size = max(1, args.batch_size)
    from dataloaders.dataloader import Dataloader
    for driver in args.images_paths:
        saved_args.val_split_path = driver
        loader = Dataloader(saved_args.dataloader).get_dataloader(saved_args, part='val', phase='val')
        out = (args.destination / string_to_valid_filename(driver)).with_suffix('.npy')
        out.parent.mkdir(parents=True, exist_ok=True)
        frames = []
        for data_dict, _ in loader:
            utils.dict_to_device(data_dict, device)
            frames.append(drive_frames(embedder, generator, data_dict))       # uint8 on the device: nothing waits for the GPU here
        video = torch.cat(frames).cpu().numpy()                             # one device-to-host transfer (and sync) per sequence
        np.save(out, video)
        logger.info(f'wrote {len(video)} frames to {out}')


if __name__ == '__main__':
    main()
