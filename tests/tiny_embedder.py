"""A two-parameter stand-in for the embedder plugin (same interface as embedders/unsupervised_pose_separate_embResNeXt_segmentation.py:
``Wrapper.get_args/get_net``, ``get_identity_embedding``, ``get_pose_embedding``, ``enable_finetuning``) used ONLY by the checkpoint
fixture: the real ResNeXt-50 + MobileNetV2 state would make the reference-written fixture 107 MB.  Registered by tests as the
plugin module ``embedders.tiny_for_tests``."""
import sys
import types

from torch import nn


class Embedder(nn.Module):
    def __init__(self, identity_embedding_size, pose_embedding_size):
        super().__init__()
        self.identity_encoder = nn.Linear(3, identity_embedding_size)
        self.pose_encoder = nn.Linear(3, pose_embedding_size)
        self.finetuning = False

    def enable_finetuning(self, data_dict=None):
        self.finetuning = True

    def get_identity_embedding(self, data_dict):
        frames = data_dict['enc_rgbs']                                   # B x K x 3 x H x W
        per_frame = self.identity_encoder(frames.mean(dim=(3, 4)))
        data_dict['embeds'] = per_frame.mean(1)
        data_dict['embeds_elemwise'] = per_frame

    def get_pose_embedding(self, data_dict):
        data_dict['pose_embedding'] = self.pose_encoder(data_dict['pose_input_rgbs'][:, 0].mean(dim=(2, 3)))

    def forward(self, data_dict):
        if not self.finetuning:
            self.get_identity_embedding(data_dict)
        self.get_pose_embedding(data_dict)


class Wrapper:
    @staticmethod
    def get_args(parser):
        pass

    @staticmethod
    def get_net(args):
        return Embedder(args.embed_channels, args.pose_embedding_size).to(args.device)


def register():
    """make ``importlib.import_module('embedders.tiny_for_tests')`` resolve to this module (plugin loaders import by name)"""
    import embedders          # the package's (or the reference's) plugin namespace must be importable already
    mod = types.ModuleType('embedders.tiny_for_tests')
    mod.Wrapper, mod.Embedder = Wrapper, Embedder
    sys.modules['embedders.tiny_for_tests'] = mod
    embedders.tiny_for_tests = mod
