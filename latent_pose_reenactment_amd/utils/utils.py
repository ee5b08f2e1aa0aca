"""Config/plugin glue with the reference's semantics (utils/utils.py:16-398): seeds, 4-level config merge
(argparse defaults < checkpoint args < configs/<name>.yaml < command line), plugin loading by
``importlib.import_module(f'{kind}.{name}')``, Meter, checkpoint save/load.  Pure host code."""
import importlib
import logging
import os
import random
import time
from argparse import Namespace
from collections import defaultdict

import torch
import yaml


def setup(args):
    log = logging.getLogger('utils.setup')
    torch.set_num_threads(1)
    os.environ['OMP_NUM_THREADS'] = '1'
    if args.random_seed is None:
        args.random_seed = int(time.time() * 2)
    log.info(f"Random Seed: {args.random_seed}")
    random.seed(args.random_seed)
    torch.manual_seed(args.random_seed)
    if str(args.device).startswith('cuda'):
        torch.cuda.manual_seed_all(args.random_seed)


def dict_to_device(d, device):
    for k, v in d.items():
        if torch.is_tensor(v):
            d[k] = v.to(device, non_blocking=True)


def load_config_file(config_name):
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for base in ('configs', os.path.join(here, 'configs')):
        path = os.path.join(base, f'{config_name}.yaml')
        if os.path.exists(path):
            logging.getLogger('utils.load_config_file').info(f"Using config {path}")
            with open(path) as f:
                text = os.path.expandvars(f.read())        # the reference resolves ${ENV} through yamlenv
            return yaml.safe_load(text) or {}
    raise FileNotFoundError(f'configs/{config_name}.yaml')


def load_module(module_type, module_name):
    return importlib.import_module(f'{module_type}.{module_name}')


def load_wrappers_for_module_list(module_name_list: str, parent_module: str):
    names = [n.strip() for n in module_name_list.split(',') if n.strip()]
    return [importlib.import_module(f'{parent_module}.{n}').Wrapper for n in names]


def torch_load(path):
    """checkpoints hold an argparse.Namespace with pathlib paths: needs weights_only=False on torch >= 2.6"""
    return torch.load(path, map_location='cpu', weights_only=False)


def get_args_and_modules(parser, use_checkpoint_args=True, custom_args={}):
    """-> (args, default_args, modules, checkpoint_object); later sources win:
    argparse defaults < checkpoint ``args`` < configs/<config_name>.yaml < ``custom_args`` < command line."""
    log = logging.getLogger('utils.get_args_and_modules')
    parser.set_defaults(**custom_args)
    args, _ = parser.parse_known_args()
    config_args = {}
    if args.config_name == '':
        log.warning("Not using any .yaml config file")
    else:
        try:
            config_args = load_config_file(args.config_name)
        except FileNotFoundError:
            log.warning(f"Could not load config {args.config_name}")
    parser.set_defaults(**config_args)
    parser.set_defaults(**custom_args)
    args, _ = parser.parse_known_args()

    checkpoint_object, checkpoint_args = None, {}
    if use_checkpoint_args and args.checkpoint_path:
        log.info(f"Loading checkpoint file {args.checkpoint_path}")
        checkpoint_object = torch_load(args.checkpoint_path)
        checkpoint_args = vars(checkpoint_object['args'])

    def resolve():
        parser.set_defaults(**checkpoint_args)
        parser.set_defaults(**config_args)
        parser.set_defaults(**custom_args)

    resolve()
    args, _ = parser.parse_known_args()          # module names are known now; let each plugin extend the parser
    m = {}
    m['generator'] = load_module('generators', args.generator).Wrapper
    m['generator'].get_args(parser)
    m['embedder'] = load_module('embedders', args.embedder).Wrapper
    m['embedder'].get_args(parser)
    m['runner'] = load_module('runners', args.runner)
    m['runner'].get_args(parser)
    m['discriminator'] = load_module('discriminators', args.discriminator).Wrapper
    m['discriminator'].get_args(parser)
    m['criterion_list'] = load_wrappers_for_module_list(args.criterions, 'criterions')
    for crit in m['criterion_list']:
        crit.get_args(parser)
    m['metric_list'] = load_wrappers_for_module_list(args.metrics, 'metrics')
    for metric in m['metric_list']:
        metric.get_args(parser)
    m['dataloader'] = load_module('dataloaders', 'dataloader').Dataloader(args.dataloader)
    m['dataloader'].get_args(parser)
    resolve()
    args, default_args = parser.parse_args(), parser.parse_args([])
    if not args.experiment_name:
        args.experiment_name = args.config_name
    return args, default_args, m, checkpoint_object


class Meter:
    """running sums / counts / last values of named scalars (utils/utils.py:196-248); NaNs are counted as no sample"""

    def __init__(self):
        self.sum = defaultdict(float)
        self.num_measurements = defaultdict(int)
        self.last_value = {}

    def add(self, name, value, num_measurements=1):
        assert num_measurements >= 0
        if num_measurements == 0:
            return
        value = float(value)
        if value == value:
            self.sum[name] += value * num_measurements
            self.num_measurements[name] += num_measurements
        else:
            self.sum[name] += 0
            self.num_measurements[name] += 0
        self.last_value[name] = value

    def keys(self):
        return self.sum.keys()

    def get_average(self, name):
        return self.sum[name] / max(1, self.num_measurements[name])

    def get_last(self, name):
        return self.last_value[name]

    def get_num_measurements(self, name):
        return self.num_measurements[name]

    def __iadd__(self, other):
        for name in other.sum:
            self.add(name, other.get_average(name), other.get_num_measurements(name))
            self.last_value[name] = other.last_value[name]
        return self


def save_model(training_module, optimizer_G, optimizer_D, args):
    """one .pth = {embedder, generator, discriminator, optimizer_G, optimizer_D, running_averages, args}, named
    model_{iteration:08}.pth under experiment_dir/checkpoints (utils/utils.py:251-295).  Rank 0 only."""
    log = logging.getLogger('utils.save_model')
    if args.rank != 0:
        return None
    tm = getattr(training_module, 'module', training_module)
    save_dict = {}
    for name in ('embedder', 'generator', 'discriminator'):
        if getattr(tm, name) is not None:
            save_dict[name] = getattr(tm, name).state_dict()
    if optimizer_G is not None:
        save_dict['optimizer_G'] = optimizer_G.state_dict()
    if optimizer_D is not None:
        save_dict['optimizer_D'] = optimizer_D.state_dict()
    if tm.running_averages is not None:
        save_dict['running_averages'] = {k: v.state_dict() for k, v in tm.running_averages.items()}
    save_dict['args'] = args
    stem = f'{args.iteration:08}'
    ckpt_dir = os.path.join(str(args.experiment_dir), 'checkpoints')
    os.makedirs(ckpt_dir, exist_ok=True)
    path = os.path.join(ckpt_dir, f'model_{stem}.pth')
    while os.path.exists(path):
        stem += '_0'
        path = os.path.join(ckpt_dir, f'model_{stem}.pth')
    try:
        torch.save(save_dict, path, pickle_protocol=-1)
        log.info(f"Saved checkpoint {path}")
        return path
    except (RuntimeError, OSError) as err:
        # the reference's guard (utils/utils.py:286-295) is for a full disk: do not leave a truncated file behind.  Anything else
        # (bad path, name too long, permissions) is a bug of the caller and must surface.
        import errno
        disk_full = isinstance(err, RuntimeError) or getattr(err, 'errno', None) in (errno.ENOSPC, errno.EDQUOT)
        if not disk_full:
            raise
        log.error(f"Could not write to {path}: {err}; removing that file")
        try:
            os.remove(path)
        except OSError:
            pass
        return None


def load_model_from_checkpoint(checkpoint_object, args=Namespace()):
    """Rebuild embedder/generator/discriminator from a checkpoint (utils/utils.py:298-398): each module is built twice
    (current args, saved args), ``enable_finetuning()`` makes the structures match, weights are copied unless that
    plugin's name changed, optimizers are restored unless switching into fine-tuning or ``args.inference``."""
    log = logging.getLogger('utils.load_model_from_checkpoint')
    saved_args = checkpoint_object['args']
    device_backup, saved_args.device = saved_args.device, 'cpu'
    finetune = bool(getattr(args, 'finetune', False))
    already_finetuned = bool(getattr(saved_args, 'finetune', False))
    assert not (already_finetuned and hasattr(args, 'finetune') and not finetune), \
        "NYI: using fine-tuned checkpoint for meta-learning"
    differing = [k for k, v in vars(args).items() if k in vars(saved_args) and v != getattr(saved_args, k)]
    running_averages = checkpoint_object.get('running_averages', {})
    modules = {}
    for name in ('embedder', 'generator', 'discriminator'):
        wrapper = load_module(f'{name}s', getattr(args, name)).Wrapper
        module, module_old = wrapper.get_net(args), wrapper.get_net(saved_args)
        if already_finetuned:
            module_old.enable_finetuning()
        module_old.load_state_dict(checkpoint_object[name])
        if finetune:
            module.enable_finetuning()
            if not already_finetuned:
                module_old.enable_finetuning()
        if name in differing:
            log.warning(f"{name} has changed in config, so not loading weights")
        else:
            module.load_state_dict(module_old.state_dict())
        modules[name] = module
    if getattr(args, 'inference', False):
        optimizer_G = optimizer_D = None
    else:
        optimizer_D = load_module('discriminators', args.discriminator).Wrapper.get_optimizer(modules['discriminator'], args)
        if 'discriminator' in differing or optimizer_D is None or (finetune and not already_finetuned):
            log.warning("Discriminator has changed in config (maybe due to finetuning), so not loading `optimizer_D`")
        else:
            optimizer_D.load_state_dict(checkpoint_object['optimizer_D'])
        runner = load_module('runners', args.runner)
        optimizer_G = runner.get_optimizer(modules['embedder'], modules['generator'], args)
        if 'generator' in differing or 'embedder' in differing or (finetune and not already_finetuned):
            log.warning("Embedder or generator has changed in config, so not loading `optimizer_G`")
        else:
            optimizer_G.load_state_dict(checkpoint_object['optimizer_G'])
    saved_args.device = device_backup
    return modules['embedder'], modules['generator'], modules['discriminator'], running_averages, saved_args, optimizer_G, optimizer_D
