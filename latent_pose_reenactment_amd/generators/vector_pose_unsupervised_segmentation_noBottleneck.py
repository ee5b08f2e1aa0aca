"""Generator plugin (reference API: generators/vector_pose_unsupervised_segmentation_noBottleneck.py:8-29).
``Wrapper.get_args`` adds the same flags with the same defaults; ``Wrapper.get_net`` returns the HIP-backed
``latent_pose_reenactment_amd.nn.Generator`` whose state_dict is key-compatible with the reference's."""
from latent_pose_reenactment_amd.nn import Generator  # noqa: F401  (re-exported: checkpoints pickle only tensors)


class Wrapper:
    @staticmethod
    def get_args(parser):
        parser.add('--gen_constant_input_size', type=int, default=4)
        parser.add('--gen_num_residual_blocks', type=int, default=2)
        parser.add('--gen_padding', type=str, default='zero', help='zero|reflection')
        parser.add('--norm_layer', type=str, default='in')

    @staticmethod
    def get_net(args):
        const_size = getattr(args, 'gen_constant_input_size', 4)      # absent from old checkpoints' args
        net = Generator(args.gen_padding, args.in_channels, args.out_channels + 1, args.num_channels, args.max_num_channels,
                        args.embed_channels, args.pose_embedding_size, args.norm_layer, const_size,
                        args.gen_num_residual_blocks, args.image_size)
        return net.to(args.device)
