"""CPU oracle for the latent-pose hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain functional PyTorch-CPU code (fp32 by default, fp64 on request), the arithmetic of the
reference's generator / discriminator / criterion hot path so that the HIP kernels can be checked against it.
It is *never* imported by the product package: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may use it.  Parity status: PINNED -- every function below is checked in
``tests/test_oracle_golden.py`` against fixtures in ``tests/golden/*.npz`` that were produced by importing the real
reference modules from ``/root/reference`` (generator script: ``tests/golden/make_golden.py``).

All tensors use the reference's NCHW layout and the reference's ``state_dict`` key names, so a checkpoint of the
reference can be fed straight in.  Citations are ``file:line`` relative to the reference repository root.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]

ADAIN_EPS = 1e-4       # generators/common/blocks.py:7  (AdaptiveNorm2d eps)
SN_EPS_CONV = 1e-4     # generators/common/blocks.py:78-80 (spectral_norm(..., eps=1e-4))
SN_EPS_DEFAULT = 1e-12 # torch default; projector linears, noBottleneck.py:98,100


# ----------------------------------------------------------------------------------------------------------------
# spectral norm (legacy torch.nn.utils.spectral_norm hook semantics)
# ----------------------------------------------------------------------------------------------------------------
def _l2normalize(x: Tensor, eps: float) -> Tensor:
    return x / torch.clamp(x.norm(), min=eps)


def sn_effective_weight(sd: State, prefix: str, eps: float, train: bool) -> Tensor:
    """``W_orig / sigma`` with one in-place power iteration on ``prefix.weight_u/_v`` when ``train``.

    Follows torch/nn/utils/spectral_norm.py::compute_weight as used by generators/common/blocks.py:76-88:
    v <- normalize(W^T u), u <- normalize(W v) under no_grad, sigma = u . (W v) with u, v constants for autograd.
    """
    w = sd[prefix + '.weight_orig']
    u = sd[prefix + '.weight_u']
    v = sd[prefix + '.weight_v']
    w_mat = w.reshape(w.shape[0], -1)
    if train:
        with torch.no_grad():
            v_new = _l2normalize(torch.mv(w_mat.t(), u), eps)
            u_new = _l2normalize(torch.mv(w_mat, v_new), eps)
            v.copy_(v_new)
            u.copy_(u_new)
    # clones: the buffers are overwritten again by later passes (D runs 3 per step) while autograd still needs them
    sigma = torch.dot(u.detach().clone(), torch.mv(w_mat, v.detach().clone()))
    return w / sigma


# ----------------------------------------------------------------------------------------------------------------
# generator pieces
# ----------------------------------------------------------------------------------------------------------------
def adain(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = ADAIN_EPS) -> Tensor:
    """generators/common/blocks.py:18-26 -- InstanceNorm2d(affine=False, biased var) then per-sample scale/shift."""
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = x.var(dim=(2, 3), unbiased=False, keepdim=True)
    xhat = (x - mean) / torch.sqrt(var + eps)
    return xhat * gamma[:, :, None, None] + beta[:, :, None, None]


def upsample2(x: Tensor) -> Tensor:
    """nn.Upsample(scale_factor=2) (nearest) -- blocks.py:74-75."""
    return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


def generator_channels(num_channels: int, max_num_channels: int, image_size: int, const_size: int = 4,
                       num_res_blocks: int = 2) -> List[Tuple[int, int, bool]]:
    """(cin, cout, upsample) for every decoder ResBlock -- noBottleneck.py:60-78."""
    n_up = int(math.log2(image_size / const_size))
    nonclamped = num_channels * (2 ** n_up)
    cur = min(nonclamped, max_num_channels)
    blocks = [(cur, cur, False)] * num_res_blocks
    for _ in range(n_up):
        cin = cur
        nonclamped //= 2
        cur = min(nonclamped, max_num_channels)
        blocks.append((cin, cur, True))
    return blocks


def split_affine_params(affine: Tensor, blocks: List[Tuple[int, int, bool]]) -> List[Tuple[Tensor, Tensor]]:
    """noBottleneck.py:108-125: per AdaptiveNorm2d in modules() order take C values as *bias*, next C as *weight*.

    Returns [(gamma, beta), ...]: norm0, norm1 of each block, then the head AdaIN."""
    out = []
    off = 0
    chans = []
    for cin, cout, _ in blocks:
        chans += [cin, cout]
    chans.append(blocks[-1][1])
    for c in chans:
        beta = affine[:, off:off + c]
        gamma = affine[:, off + c:off + 2 * c]
        out.append((gamma, beta))
        off += 2 * c
    assert off == affine.shape[1], (off, affine.shape)
    return out


RELU_REPLAY: Optional[List[Tensor]] = None      # tie-masked checks of the critic / VGG stacks: activation patterns, consumed in site order


def _relu(x: Tensor) -> Tensor:
    """ReLU site of the discriminator / perceptual stacks: the true ReLU, or -- when ``RELU_REPLAY`` holds the activation patterns
    recorded by the implementation under test -- that implementation's own piecewise-linear branch (see ``_relu_m``)."""
    if RELU_REPLAY is None:
        return torch.relu(x)
    m = RELU_REPLAY.pop(0)
    assert m.shape == x.shape, (m.shape, x.shape)
    return x * m.to(x.dtype)


def _l1(a: Tensor, b: Tensor) -> Tensor:
    """F.l1_loss site of the perceptual stacks: mean|a - b|, or -- under ``RELU_REPLAY`` -- the same piecewise-linear branch with the
    recorded sign pattern s of (a - b): mean(s * (a - b)).  sign() is the other discontinuity of this path: where two features
    agree to within rounding the derivative +-1/numel flips between two correct implementations."""
    if RELU_REPLAY is None:
        return F.l1_loss(a, b)
    sgn = RELU_REPLAY.pop(0)
    assert sgn.shape == a.shape, (sgn.shape, a.shape)
    return (sgn.to(a.dtype) * (a - b)).mean()


def _relu_m(x: Tensor, mask: Optional[Tensor]) -> Tensor:
    """ReLU, or -- tie-masked parity checks -- the same piecewise-linear branch with a PRESCRIBED activation pattern ``mask`` (the
    pattern of the implementation under test): pre-activations within rounding distance of 0 flip between two correct
    implementations, and a flipped unit changes a gradient by a whole term, so gradients are compared on the tested
    implementation's own branch.  Forward outputs are always compared with the true ReLU."""
    return torch.relu(x) if mask is None else x * mask.to(x.dtype)


def conv3x3(x: Tensor, w: Tensor, b: Optional[Tensor], padding: str = 'zero') -> Tensor:
    """the 3x3 conv of a ResBlock behind its padding layer (blocks.py:76-88): `nn.Sequential()` + Conv2d(padding=1) for nn.ZeroPad2d,
    nn.ReflectionPad2d(1) + Conv2d(padding=0) for --gen_padding / --dis_padding reflection"""
    if padding == 'reflection':
        return F.conv2d(F.pad(x, (1, 1, 1, 1), mode='reflect'), w, b, 1, 0)
    assert padding == 'zero', padding
    return F.conv2d(x, w, b, 1, 1)


def resblock_ada(x: Tensor, sd: State, prefix: str, aff0, aff1, upsample: bool, train: bool, masks=(None, None), padding: str = 'zero') -> Tensor:
    """blocks.ResBlock with norm_layer='adain' (blocks.py:47-111): pre-activation, convs without bias, SN eps 1e-4."""
    i1, i2 = (4, 8) if upsample else (3, 7)
    h = _relu_m(adain(x, *aff0), masks[0])
    if upsample:
        h = upsample2(h)
    w1 = sn_effective_weight(sd, f'{prefix}.block.{i1}', SN_EPS_CONV, train)
    h = conv3x3(h, w1, None, padding)
    h = _relu_m(adain(h, *aff1), masks[1])
    w2 = sn_effective_weight(sd, f'{prefix}.block.{i2}', SN_EPS_CONV, train)
    h = conv3x3(h, w2, None, padding)
    if f'{prefix}.skip.1.weight_orig' in sd:   # in != out or upsample (blocks.py:92-103); Upsample is skip.0
        s = upsample2(x) if upsample else x
        ws = sn_effective_weight(sd, f'{prefix}.skip.1', SN_EPS_CONV, train)
        s = F.conv2d(s, ws, sd[f'{prefix}.skip.1.bias'])
        return h + s
    if f'{prefix}.skip.0.weight_orig' in sd:   # in != out without upsample
        ws = sn_effective_weight(sd, f'{prefix}.skip.0', SN_EPS_CONV, train)
        return h + F.conv2d(x, ws, sd[f'{prefix}.skip.0.bias'])
    return h + x


def generator_forward(sd: State, identity: Tensor, pose: Tensor, *, num_channels: int, max_num_channels: int,
                      image_size: int, train: bool, const_size: int = 4, num_res_blocks: int = 2,
                      relu_masks: Optional[List[Tensor]] = None, fsth_plus: bool = False, padding: str = 'zero') -> Tuple[Tensor, Tensor]:
    """Generator.forward (noBottleneck.py:165-181).  ``identity`` is data_dict['embeds'] (B x E) or the finetuned
    ``identity_embedding`` (1 x E, expanded).  Returns (fake_rgbs, fake_segm).  SN buffers in ``sd`` are updated in
    place when ``train``.  ``relu_masks``: prescribed activation patterns of the 17 AdaIN+ReLU sites in execution order
    (tie-masked gradient checks, see ``_relu_m``).  ``fsth_plus``: generators/FSTH_plus.py -- ``pose`` is then
    ``dec_keypoints[:, 0] - 0.5`` and the projector three plain Linear layers with LeakyReLU(0.05) (FSTH_plus.py:96-103,129-139)."""
    b = pose.shape[0]
    if identity.shape[0] == 1 and b != 1:
        identity = identity.expand(b, -1)
    joint = torch.cat((identity, pose), dim=1)
    if fsth_plus:
        h = F.leaky_relu(F.linear(joint, sd['affine_params_projector.0.weight'], sd['affine_params_projector.0.bias']), 0.05)
        h = F.leaky_relu(F.linear(h, sd['affine_params_projector.2.weight'], sd['affine_params_projector.2.bias']), 0.05)
        affine = F.linear(h, sd['affine_params_projector.4.weight'], sd['affine_params_projector.4.bias'])
    else:
        # affine_params_projector: SN-Linear -> ReLU -> SN-Linear (noBottleneck.py:96-101), default SN eps
        w0 = sn_effective_weight(sd, 'affine_params_projector.0', SN_EPS_DEFAULT, train)
        h = torch.relu(F.linear(joint, w0, sd['affine_params_projector.0.bias']))
        w2 = sn_effective_weight(sd, 'affine_params_projector.2', SN_EPS_DEFAULT, train)
        affine = F.linear(h, w2, sd['affine_params_projector.2.bias'])

    blocks = generator_channels(num_channels, max_num_channels, image_size, const_size, num_res_blocks)
    affs = split_affine_params(affine, blocks)
    x = sd['constant.constant'].expand(b, -1, -1, -1)
    # NOTE on SN ordering: the hook fires per module at its forward, i.e. in execution order; power iterations of
    # distinct layers are independent so the order does not matter numerically.
    for i, (cin, cout, up) in enumerate(blocks):
        mk = (None, None) if relu_masks is None else (relu_masks[2 * i], relu_masks[2 * i + 1])
        x = resblock_ada(x, sd, f'decoder_blocks.{i}', affs[2 * i], affs[2 * i + 1], up, train, mk, padding)     # (the head conv below keeps zero padding: noBottleneck.py:80-88)
    nb = len(blocks)
    x = _relu_m(adain(x, *affs[2 * nb]), None if relu_masks is None else relu_masks[2 * nb])
    wh = sn_effective_weight(sd, f'decoder_blocks.{nb + 2}', SN_EPS_CONV, train)
    x = torch.tanh(F.conv2d(x, wh, sd[f'decoder_blocks.{nb + 2}.bias'], 1, 1))
    rgb = x[:, :-1] * 0.75 + 0.5
    segm = x[:, -1:] * 0.5 + 0.5
    return rgb * segm, segm


# ----------------------------------------------------------------------------------------------------------------
# discriminator (discriminators/no_landmarks.py)
# ----------------------------------------------------------------------------------------------------------------
def resblock_none(x_relu: Tensor, sd: State, prefix: str, downsample: bool, train: bool, padding: str = 'zero') -> Tensor:
    """blocks.ResBlock with norm_layer='none' as the reference actually behaves: its first layer is
    ReLU(inplace=True) on the block *input* (blocks.py:71-73), so block, skip and identity all see relu(x)
    (SURVEY Appendix B).  ``x_relu`` must already be relu(x)."""
    w1 = sn_effective_weight(sd, f'{prefix}.block.2', SN_EPS_CONV, train)
    h = conv3x3(x_relu, w1, sd[f'{prefix}.block.2.bias'], padding)
    h = _relu(h)
    w2 = sn_effective_weight(sd, f'{prefix}.block.5', SN_EPS_CONV, train)
    h = conv3x3(h, w2, sd[f'{prefix}.block.5.bias'], padding)
    if downsample:
        h = F.avg_pool2d(h, 2)
    if f'{prefix}.skip.0.weight_orig' in sd:
        ws = sn_effective_weight(sd, f'{prefix}.skip.0', SN_EPS_CONV, train)
        s = F.conv2d(x_relu, ws, sd[f'{prefix}.skip.0.bias'])
        if downsample:
            s = F.avg_pool2d(s, 2)
        return h + s
    return h + x_relu


def discriminator_layout(image_size: int, dis_num_blocks: int) -> List[bool]:
    """downsample flag of each entry of ``blocks`` -- no_landmarks.py:69-78."""
    num_down = min(int(math.log(image_size, 2)) - 2, dis_num_blocks)
    return [True] * (num_down - 1) + [False] * (dis_num_blocks - num_down)


def discriminator_pass(sd: State, x: Tensor, embed: Optional[Tensor], *, image_size: int, dis_num_blocks: int,
                       train: bool, padding: str = 'zero') -> Tuple[Tensor, List[Tensor]]:
    """Discriminator.pass_inputs (no_landmarks.py:90-108).  Returned features are what the reference's list holds
    *after* the call: feats[0..n-2] post-ReLU (mutated in place by the next block), feats[n-1] pre-ReLU."""
    w = sn_effective_weight(sd, 'down_block.0', SN_EPS_CONV, train)
    h = _relu(F.conv2d(x, w, sd['down_block.0.bias'], 1, 1))
    w = sn_effective_weight(sd, 'down_block.2', SN_EPS_CONV, train)
    h = F.avg_pool2d(F.conv2d(h, w, sd['down_block.2.bias'], 1, 1), 2)
    w = sn_effective_weight(sd, 'skip.0', SN_EPS_CONV, train)
    s = F.avg_pool2d(F.conv2d(x, w, sd['skip.0.bias']), 2)
    out = h + s
    feats = []
    for i, down in enumerate(discriminator_layout(image_size, dis_num_blocks)):
        out_relu = _relu(out)
        feats.append(out_relu)
        out = resblock_none(out_relu, sd, f'blocks.{i}', down, train, padding)     # (the stem above keeps zero padding: no_landmarks.py:52-60)
    feats.append(out)
    h = _relu(out)
    h = h.reshape(h.shape[0], h.shape[1], -1).sum(2)
    w = sn_effective_weight(sd, 'linear', SN_EPS_CONV, train)
    lin = F.linear(h, w, sd['linear.bias'])[:, 0]
    score = (h * embed).sum(1) + lin if embed is not None else lin
    return score, feats


def discriminator_forward(sd: State, fake_rgbs: Tensor, target_rgbs: Tensor, label: Tensor, *, image_size: int,
                          dis_num_blocks: int, train: bool, embed_eps: float = SN_EPS_CONV, padding: str = 'zero') -> Dict[str, object]:
    """Discriminator.forward (no_landmarks.py:138-166): embedding lookup through SN, then three passes
    (fake -> G, fake.detach -> D, real), each running its own power iteration in train mode."""
    w_embed = sn_effective_weight(sd, 'embed', embed_eps, train)
    embed = w_embed[label]
    fake_score_G, fake_features = discriminator_pass(sd, fake_rgbs, embed, image_size=image_size,
                                                     dis_num_blocks=dis_num_blocks, train=train, padding=padding)
    fake_score_D, _ = discriminator_pass(sd, fake_rgbs.detach(), embed.detach(), image_size=image_size,
                                         dis_num_blocks=dis_num_blocks, train=train, padding=padding)
    real_score, real_features = discriminator_pass(sd, target_rgbs, embed, image_size=image_size,
                                                   dis_num_blocks=dis_num_blocks, train=train, padding=padding)
    return dict(fake_features=fake_features, real_features=real_features, real_embedding=embed,
                fake_score_G=fake_score_G, fake_score_D=fake_score_D, real_score=real_score)


# ----------------------------------------------------------------------------------------------------------------
# criterions
# ----------------------------------------------------------------------------------------------------------------
def adversarial_gan(fake_score_G: Tensor, fake_score_D: Tensor, real_score: Tensor) -> Tuple[Tensor, Tensor]:
    """criterions/adversarial.py:34-57 with gan_type='gan' -> (loss_G, loss_D)."""
    loss_D = torch.relu(1. - real_score).mean() + torch.relu(1. + fake_score_D).mean()
    return -fake_score_G.mean(), loss_D


def feature_matching(fake_feats: List[Tensor], real_feats: List[Tensor], fm_weight: float = 10.0) -> Tensor:
    """criterions/featmat.py:16-27."""
    return sum(F.l1_loss(f, r.detach()) for f, r in zip(fake_feats, real_feats)) / len(fake_feats) * fm_weight


def dice(fake_segm: Tensor, real_segm: Tensor, dice_weight: float = 1.0) -> Tensor:
    """criterions/dice.py:20-39 including its B x 1 vs B x 3 broadcast."""
    if fake_segm.dim() > 4:
        fake_segm = fake_segm[:, 0]
    if real_segm.dim() > 4:
        real_segm = real_segm[:, 0]
    numer = (2 * fake_segm * real_segm).sum()
    denom = (fake_segm ** 2).sum() + (real_segm ** 2).sum()
    return -torch.log(numer / denom) * dice_weight


def dis_embed(embeds_elemwise: Tensor, real_embedding: Tensor, weight: float = 1e-2) -> Tensor:
    """criterions/dis_embed.py:21-34."""
    fe = embeds_elemwise[:, 0] if embeds_elemwise.dim() > 2 else embeds_elemwise
    return F.l1_loss(fe, real_embedding.detach()) * weight


VGG19_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']
VGG16_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']
CAFFE_MEAN = (103.939, 116.779, 123.680)


def vgg_layer_plan(cfg, num_layers: int = 30):
    """torchvision ``features`` layout (conv, relu, ..., pool) truncated to the first ``num_layers`` modules, with
    MaxPool -> AvgPool as in criterions/common/perceptual_loss.py:70-86.  Returns [('conv', idx, cout)|('relu',)|('pool',)]."""
    plan = []
    idx = 0
    for v in cfg:
        if v == 'M':
            plan.append(('pool',))
            idx += 1
        else:
            plan.append(('conv', idx, v))
            plan.append(('relu',))
            idx += 2
    return plan[:num_layers]


def perceptual_loss(sd: State, fake: Tensor, real: Tensor, weight: float, cfg, num_layers: int = 30,
                    key_prefix: str = '') -> Tensor:
    """PerceptualLoss.forward for net in {'caffe','face'} (criterions/common/perceptual_loss.py:91-110):
    x<-(x+1)/2, (x - mean_bgr/255)*255 on RGB-ordered input, L1 at every ReLU, both images forwarded."""
    mean = torch.tensor(CAFFE_MEAN, dtype=fake.dtype) / 255.
    std = torch.tensor([1., 1., 1.], dtype=fake.dtype) / 255.
    fi = ((fake + 1) / 2 - mean[None, :, None, None]) / std[None, :, None, None]
    ft = ((real.detach() + 1) / 2 - mean[None, :, None, None]) / std[None, :, None, None]
    loss = 0
    for item in vgg_layer_plan(cfg, num_layers):
        if item[0] == 'conv':
            w, bb = sd[f'{key_prefix}{item[1]}.weight'], sd[f'{key_prefix}{item[1]}.bias']
            fi, ft = F.conv2d(fi, w, bb, 1, 1), F.conv2d(ft, w, bb, 1, 1)
        elif item[0] == 'relu':
            fi, ft = _relu(fi), torch.relu(ft)       # (only the fake branch carries gradient)
            loss = loss + _l1(fi, ft)
        else:
            fi, ft = F.avg_pool2d(fi, 2, 2), F.avg_pool2d(ft, 2, 2)
    return loss * weight


def crop_and_resize_fixed(images: Tensor, crop_factor: float = 1 / 1.8) -> Tensor:
    """criterions/idt_embed.py:37-49 + crop_and_resize (:58-83): fixed centre bbox, affine_grid(align_corners=False)
    + grid_sample(bilinear, reflection) back to the input size."""
    b, c, h, w = images.shape
    t = h * (1 - crop_factor) / 2
    l = w * (1 - crop_factor) / 2
    bb, r = h - t, w - l
    theta = torch.zeros(b, 2, 3, dtype=torch.float32)
    theta[:, 0, 0] = (r - l) / w
    theta[:, 1, 1] = (bb - t) / h
    theta[:, 0, 2] = (l + r) / w - 1
    theta[:, 1, 2] = (t + bb) / h - 1
    grid = F.affine_grid(theta.to(images.dtype), (b, c, h, w), align_corners=False)
    return F.grid_sample(images, grid, mode='bilinear', padding_mode='reflection', align_corners=False)


# ----------------------------------------------------------------------------------------------------------------
# optimizers / EMA (runners/holycow.py:99-109; utils/radam.py:29-95; torch.optim.Adam)
# ----------------------------------------------------------------------------------------------------------------
def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1: float, beta2: float,
              eps: float) -> None:
    """torch.optim.Adam (no amsgrad, no weight decay) single-tensor update, in place; ``step`` is 1-based."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def radam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1: float, beta2: float,
               eps: float) -> None:
    """utils/radam.py:29-95 (degenerated_to_sgd=True, weight_decay=0), in place; ``step`` is 1-based."""
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    beta2_t = beta2 ** step
    n_max = 2 / (1 - beta2) - 1
    n_sma = n_max - 2 * step * beta2_t / (1 - beta2_t)
    if n_sma >= 5:
        step_size = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_max - 4) * (n_sma - 2) / n_sma * n_max / (n_max - 2)) \
            / (1 - beta1 ** step)
        p.addcdiv_(m, v.sqrt().add_(eps), value=-step_size * lr)
    else:
        step_size = 1.0 / (1 - beta1 ** step)
        p.add_(m, alpha=-step_size * lr)


def ema_update(avg: Tensor, cur: Tensor, alpha: float) -> None:
    """runners/holycow.py:104-106."""
    avg.mul_(alpha).add_(cur * (1 - alpha))
