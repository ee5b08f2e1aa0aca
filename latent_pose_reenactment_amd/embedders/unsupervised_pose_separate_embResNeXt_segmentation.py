"""Embedder plugin (reference API: embedders/unsupervised_pose_separate_embResNeXt_segmentation.py:7-63):
identity = ResNeXt-50 32x4d over the B x K encoder frames averaged over K, pose = MobileNetV2 on one frame."""
from torch import nn

from .backbones import mobilenet_v2, resnext50_32x4d


class Wrapper:
    @staticmethod
    def get_args(parser):
        parser.add('--average_function', type=str, default='sum', help='sum|max')

    @staticmethod
    def get_net(args):
        return Embedder(args.embed_channels, args.pose_embedding_size, args.average_function).to(args.device)


class Embedder(nn.Module):
    def __init__(self, identity_embedding_size, pose_embedding_size, average_function):
        super().__init__()
        self.identity_embedding_size = identity_embedding_size
        self.pose_embedding_size = pose_embedding_size
        self.identity_encoder = resnext50_32x4d(num_classes=identity_embedding_size)
        self.pose_encoder = mobilenet_v2(num_classes=pose_embedding_size)
        if average_function not in ('sum', 'max'):
            raise ValueError("Incorrect `average_function` argument, expected `sum` or `max`")
        self.average_function = average_function
        self.finetuning = False

    def enable_finetuning(self, data_dict=None):
        self.finetuning = True

    def get_identity_embedding(self, data_dict):
        frames = data_dict['enc_rgbs']
        b, k, c, h, w = frames.shape
        per_frame = self.identity_encoder(frames.reshape(b * k, c, h, w)).view(b, k, -1)
        assert per_frame.shape[2] == self.identity_embedding_size
        data_dict['embeds'] = per_frame.mean(1) if self.average_function == 'sum' else per_frame.max(1)[0]
        data_dict['embeds_elemwise'] = per_frame

    def get_pose_embedding(self, data_dict):
        data_dict['pose_embedding'] = self.pose_encoder(data_dict['pose_input_rgbs'][:, 0])

    def forward(self, data_dict):
        from latent_pose_reenactment_amd import streams
        if not self.finetuning and streams.enabled(data_dict['pose_input_rgbs'], 'encoders'):
            # the two encoders are independent: the pose encoder's ~600 short launches (forward and, through autograd's per-node streams,
            # backward) run beside the identity encoder's large ones instead of after them
            with streams.branch(data_dict['pose_input_rgbs'].device, 0) as b:
                self.get_pose_embedding(data_dict)
            self.get_identity_embedding(data_dict)
            b.join(data_dict['pose_embedding'])
            return
        if not self.finetuning:          # after fine-tuning the identity lives in the generator (train.py:263-266)
            self.get_identity_embedding(data_dict)
        self.get_pose_embedding(data_dict)
