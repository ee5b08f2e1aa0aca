#!/usr/bin/env python3
"""Training entry point with the reference's CLI and flow (train.py:22-310): same flags, same 4-level config merge, same
plugin loading, fine-tuning bootstrap, epoch loop and SIGINT/SIGTERM checkpointing.  Differences, all on purpose:
  * data-parallel training uses ``parallel.GradReducer`` (RCCL over xGMI) instead of apex.Reducer; the launcher contract is
    unchanged (``python -m torch.distributed.launch/run --nproc_per_node=N train.py ...``; ``--local_rank`` or LOCAL_RANK);
  * the horovod (>8 GPU) branch is not provided (broken upstream); TensorBoard/visual logging is out of scope.
Run from this directory (or with it on PYTHONPATH) so that ``generators.<name>`` etc. resolve to these plugins."""
import os
import sys

os.environ.setdefault('OMP_NUM_THREADS', '1')
HERE = os.path.dirname(os.path.abspath(__file__))
for _p in (HERE, os.path.dirname(HERE)):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import logging  # noqa: E402
from pathlib import Path  # noqa: E402

import torch  # noqa: E402

from utils import utils  # noqa: E402
from utils.argparse_utils import MyArgumentParser  # noqa: E402
from utils.utils import get_args_and_modules, load_model_from_checkpoint, save_model, setup  # noqa: E402

logging.basicConfig(level=logging.INFO, stream=sys.stdout,
                    format="PID %(process)d - %(asctime)s - %(levelname)s - %(name)s - %(message)s")
logger = logging.getLogger('train.py')


def build_parser():
    parser = MyArgumentParser(conflict_handler='resolve')
    parser.add('--config_name', type=str, default="")
    for kind in ('generator', 'embedder', 'discriminator', 'criterions', 'metrics', 'dataloader', 'runner'):
        parser.add(f'--{kind}', type=str, default="")
    parser.add('--args-to-ignore', type=str, default="checkpoint,splits_dir,experiments_dir,extension,experiment_name,rank,local_rank,world_size")
    parser.add('--experiments_dir', type=Path, default="data/experiments")
    parser.add('--experiment_name', type=str, default="")
    parser.add('--train_split_path', default="data/splits/train.csv", type=Path)
    parser.add('--val_split_path', default="data/splits/val.csv", type=Path)
    parser.add('--vgg_weights_dir', default="criterions/common/", type=str)
    parser.add('--num_epochs', type=int, default=10 ** 9)
    parser.add('--set_eval_mode_in_train', action='store_bool', default=False)
    parser.add('--set_eval_mode_in_test', action='store_bool', default=True)
    parser.add('--save_frequency', type=int, default=1, help="Save checkpoint every X epochs. If 0, save only at the end of training")
    parser.add('--logging', action='store_bool', default=True)
    parser.add('--skip_eval', action='store_bool', default=True)
    parser.add('--profile_flops', action='store_bool', default=False)
    parser.add('--weights_running_average', action='store_bool', default=True)
    parser.add('--finetune', action='store_bool', default=False)
    parser.add('--inference', action='store_bool', default=False)
    parser.add('--in_channels', type=int, default=3)
    parser.add('--out_channels', type=int, default=3)
    parser.add('--num_channels', type=int, default=64)
    parser.add('--max_num_channels', type=int, default=512)
    parser.add('--embed_channels', type=int, default=512)
    parser.add('--pose_embedding_size', type=int, default=136)
    parser.add('--image_size', type=int, default=256)
    parser.add('--optimizer', default='Adam', type=str, choices=['Adam', 'RAdam'])
    parser.add('--lr_gen', default=5e-5, type=float)
    parser.add('--beta1', default=0.0, type=float, help='beta1 for Adam')
    parser.add('--device', type=str, default='cuda')
    parser.add('--num_gpus', type=int, default=1, help='data-parallel processes on this node (RCCL), at most 8')
    parser.add('--rank', type=int, default=0, help='global rank, DO NOT SET')
    parser.add('--local_rank', type=int, default=int(os.environ.get('LOCAL_RANK', 0)), help='"rank" within a machine, DO NOT SET')
    parser.add('--world_size', type=int, default=1, help='number of devices, DO NOT SET')
    parser.add('--random_seed', type=int, default=123)
    parser.add('--checkpoint_path', type=str, default='')
    parser.add('--saver', type=str, default='')
    parser.add('--hip_graph', action='store_bool', default=False, help='replay the training step as captured hipGraphs')
    return parser


def main():
    args, default_args, m, checkpoint_object = get_args_and_modules(build_parser(), use_checkpoint_args=True)
    setup(args)
    if args.num_gpus == 1:
        args.rank = args.local_rank = 0
        args.world_size = 1
    elif 1 < args.num_gpus <= 8:
        args.rank, args.world_size = args.local_rank, args.num_gpus
        if str(args.device).startswith('cuda'):
            torch.cuda.set_device(args.local_rank)
            args.device = f'cuda:{args.local_rank}'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.distributed.init_process_group(backend='nccl' if str(args.device).startswith('cuda') else 'gloo', init_method='env://')
    else:
        raise NotImplementedError('more than 8 GPUs (the reference horovod branch) is not supported')
    logger.info(f"Initialized the process group, my rank is {args.rank}")
    if args.finetune and args.num_gpus > 1:
        if args.local_rank == 0:
            logger.warning("Sorry, multi-GPU fine-tuning is NYI, setting `--num_gpus=1`")
            args.num_gpus = 1
        else:
            logger.warning("Sorry, multi-GPU fine-tuning is NYI, shutting down all processes but one")
            sys.exit()

    dataloader_train = m['dataloader'].get_dataloader(args, part='train', phase='train')
    runner = m['runner']
    if args.checkpoint_path != "":
        if checkpoint_object is None:
            raise FileNotFoundError(f"Checkpoint `{args.checkpoint_path}` not found")
        embedder, generator, discriminator, running_averages, saved_args, optimizer_G, optimizer_D = \
            load_model_from_checkpoint(checkpoint_object, args)
    else:
        if args.finetune:
            logger.error("`--finetune` is set, but `--checkpoint_path` isn't. This has to be a mistake.")
        discriminator = m['discriminator'].get_net(args)
        generator = m['generator'].get_net(args)
        embedder = m['embedder'].get_net(args)
        running_averages = {}
        optimizer_G = runner.get_optimizer(embedder, generator, args)
        optimizer_D = m['discriminator'].get_optimizer(discriminator, args)
    criterion_list = [crit.get_net(args) for crit in m['criterion_list']]
    if not args.weights_running_average:
        running_averages = None
    args.experiment_dir = Path(args.experiments_dir) / (args.experiment_name or 'run')
    training_module = runner.TrainingModule(embedder, generator, discriminator, criterion_list, [], running_averages)

    saved = {'done': False}
    if args.rank == 0:
        import signal
        parent = os.getpid()

        def save_and_exit(*_):
            if saved['done'] or os.getpid() != parent:
                return
            saved['done'] = True
            logger.info("Interrupted, saving the current model")
            save_model(training_module, optimizer_G, optimizer_D, args)
            sys.exit()
        signal.signal(signal.SIGINT, save_and_exit)
        signal.signal(signal.SIGTERM, save_and_exit)

    def attach_reducer(broadcast):
        """data-parallel gradient exchange over the optimizers' flat gradient arenas (in place, no gather/scatter); must be rebuilt
        whenever the optimizers are (their arenas are what gets all-reduced)"""
        if 1 < args.num_gpus <= 8:
            from latent_pose_reenactment_amd.parallel import GradReducer
            training_module.reducer = GradReducer(training_module, finetune=args.finetune, broadcast=broadcast,
                                                  optimizer_G=optimizer_G, optimizer_D=optimizer_D,
                                                  max_batch=max(1, args.batch_size // args.num_gpus))
            training_module.__dict__['module'] = training_module
    attach_reducer(broadcast=True)

    if args.finetune:
        # fine-tuning bootstrap (train.py:218-279): average identity embedding over all frames of the person
        logger.info(f"For fine-tuning, computing an averaged identity embedding from {len(dataloader_train.dataset)} frames")
        training_module.eval()
        chunks = []
        with torch.no_grad():
            emb = training_module.running_averages.get('embedder', training_module.embedder) if training_module.running_averages \
                else training_module.embedder
            for data_dict, _ in dataloader_train:
                utils.dict_to_device(data_dict, args.device)
                emb.get_identity_embedding(data_dict)
                chunks.append(data_dict['embeds_elemwise'].view(-1, args.embed_channels))
            identity = torch.cat(chunks).mean(0)
        data_dict = {'embeds': identity[None]}
        training_module.generator.enable_finetuning(data_dict)
        training_module.discriminator.enable_finetuning(data_dict)
        training_module.embedder.enable_finetuning()
        if args.weights_running_average:
            training_module.running_averages['generator'].enable_finetuning(data_dict)
            training_module.running_averages['embedder'].enable_finetuning()
        else:
            training_module.initialize_running_averages(None)
        optimizer_G = runner.get_optimizer(training_module.embedder, training_module.generator, args)
        optimizer_D = m['discriminator'].get_optimizer(discriminator, args)
        attach_reducer(broadcast=False)

    logger.info("Entering training loop")
    for epoch in range(args.num_epochs):
        training_module.train(not args.set_eval_mode_in_train)
        torch.set_grad_enabled(True)
        meter = runner.run_epoch(dataloader_train, training_module, optimizer_G, optimizer_D, epoch, args, phase='train')
        if args.rank == 0 and meter is not None:
            logger.info(f"Epoch {epoch} (iteration {args.iteration}): " +
                        ", ".join(f"{k} {meter.get_last(k):.6g} (avg {meter.get_average(k):.6g})" for k in sorted(meter.keys())))
        if not args.skip_eval:
            raise NotImplementedError("NYI: validation")
        if args.rank == 0:
            last = epoch == args.num_epochs - 1
            if last or (args.save_frequency != 0 and epoch % args.save_frequency == 0):
                save_model(training_module, optimizer_G, optimizer_D, args)


if __name__ == '__main__':
    main()
