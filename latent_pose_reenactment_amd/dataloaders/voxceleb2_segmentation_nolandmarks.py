"""Placeholder for the reference's VoxCeleb2 dataset plugin (dataloaders/voxceleb2_segmentation_nolandmarks.py:182-248).

Reading VoxCeleb2 frames needs cv2 / imgaug / pandas and the dataset itself: host-side I/O that is out of scope of the MI355X
hot path.  ``configs/default.yaml`` keeps the reference's dataloader name so that the file stays a drop-in; this
module makes that name resolve and fail with a clear message instead of an ImportError.  ``synthetic_voxceleb2`` produces batches
with exactly this plugin's output contract."""
from .synthetic_voxceleb2 import Dataset as _Synthetic


class Dataset(_Synthetic):
    @staticmethod
    def get_dataset(args, part):
        raise FileNotFoundError(
            "dataloader 'voxceleb2_segmentation_nolandmarks' reads VoxCeleb2 from disk (cv2/imgaug/pandas host pipeline), which is not "
            "part of this package; pass `--dataloader synthetic_voxceleb2` for batches with the same data_dict/target_dict contract")
