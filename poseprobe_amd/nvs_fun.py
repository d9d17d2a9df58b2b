"""Full-image inference driver on the HIP path: counterpart of the reference's `render_viewpoints`
(lib/nvs_fun.py:39-188) - every pixel of every requested view through `Voxurf.inference` in chunks of 4096 rays, the
per-ray outputs assembled into images (`rgb_marched`, `disp`, `alphainv_cum`, `normal_marched`), PSNR against ground truth
(full image, foreground, background, as the reference logs them), optional files.

Same call signature and return value (`rgbs [V,H,W,3]`, `disps [V,H,W,1]` as numpy) so that `run.py`-style callers work
unchanged; the metrics of the last call are kept in `render_viewpoints.last` (the reference only prints them).  SSIM / LPIPS
need networks that cannot exist offline (lib/utils.py rgb_lpips fetches AlexNet / VGG weights): requesting them raises.
Images are written as PNG by a 20-line encoder (zlib) - imageio is not a dependency.
"""
import os
import struct
import zlib

import numpy as np
import torch

from . import camera
from . import voxurf_coarse as Model

CHUNK = 4096                        # rays per Voxurf.inference call (lib/nvs_fun.py:83)
KEYS = ('rgb_marched', 'disp', 'alphainv_cum', 'normal_marched')


def to8b(x):
    """lib/utils.py to8b."""
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)


def write_png(path, img8):
    """uint8 [H,W,3] (or [H,W] / [H,W,1] grey) -> PNG file."""
    img8 = np.asarray(img8, dtype=np.uint8)
    if img8.ndim == 3 and img8.shape[2] == 1:
        img8 = img8[:, :, 0]
    h, w = img8.shape[:2]
    colour = 2 if img8.ndim == 3 else 0
    raw = b''.join(b'\x00' + img8[r].tobytes() for r in range(h))

    def chunk(tag, data):
        body = tag + data
        return struct.pack('>I', len(data)) + body + struct.pack('>I', zlib.crc32(body) & 0xffffffff)

    with open(path, 'wb') as f:
        f.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, colour, 0, 0, 0))
                + chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


@torch.no_grad()
def render_view(model, H, W, K, c2w, ndc, render_kwargs, flip_x=False, flip_y=False, keys=KEYS, chunk=CHUNK):
    """One view: dict of [H, W, C] tensors on the model's device."""
    rays_o, rays_d, viewdirs = Model.get_rays_of_a_view(int(H), int(W), K, c2w, ndc, inverse_y=render_kwargs['inverse_y'],
                                                        flip_x=flip_x, flip_y=flip_y)
    ro, rd, vd = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), viewdirs.reshape(-1, 3)
    parts = {k: [] for k in keys}
    # only per-ray entries are kept of a chunk: the sync-free entry point (same kernels, no host round trip per chunk) serves them
    ray_level = set(keys) <= {'rgb_marched', 'disp', 'alphainv_cum', 'normal_marched', 'depth', 'cum_weights'} and hasattr(model, 'inference_rays')
    run = model.inference_rays if ray_level else model.inference
    for b in range(0, ro.shape[0], chunk):
        out = run(ro[b:b + chunk], rd[b:b + chunk], vd[b:b + chunk], training=False, **render_kwargs)
        for k in keys:
            if out.get(k) is not None:
                parts[k].append(out[k])
    return {k: torch.cat(v).reshape(int(H), int(W), -1) for k, v in parts.items() if v}


def psnr_terms(rgb, gt, mask=None):
    """(full, foreground, background) PSNR in the reference's normalisation (lib/nvs_fun.py:118-123): the masked sums run
    over all three channels but are divided by the number of masked PIXELS."""
    full = -10. * np.log10(np.mean(np.square(rgb - gt)))
    if mask is None:
        return full, 0., 0.
    back = -10. * np.log10(np.sum(np.square(rgb * (1 - mask) - gt * (1 - mask))) / np.sum(1 - mask))
    fore = -10. * np.log10(np.sum(np.square(rgb * mask - gt * mask)) / np.sum(mask))
    return full, fore, back


def render_viewpoints(model, render_poses, cfg, HW, Ks, ndc, render_kwargs, gt_imgs=None, masks=None, savedir=None,
                      render_factor=0, idx=None, eval_ssim=False, eval_lpips_alex=False, eval_lpips_vgg=False, use_bar=True,
                      step=0, rgb_only=False):
    """render_poses [V,3|4,4] world-to-camera (inverted here, as at lib/nvs_fun.py:47)."""
    if eval_ssim or eval_lpips_alex or eval_lpips_vgg:
        raise NotImplementedError('render_viewpoints: SSIM / LPIPS (lib/utils.py rgb_ssim, rgb_lpips) need network weights '
                                  'that are fetched online; compute them outside with the returned images')
    dev = next(model.parameters()).device
    poses = torch.as_tensor(np.asarray(render_poses.cpu() if isinstance(render_poses, torch.Tensor) else render_poses),
                            dtype=torch.float32)
    if poses.shape[1] == 4:
        poses = poses[:, :3, :]
    assert len(poses) == len(HW) and len(HW) == len(Ks)
    c2ws = camera.pose.invert(poses).to(dev)
    HW, Ks = np.array(HW), np.array(Ks, dtype=np.float32)
    if render_factor != 0:
        HW = HW // render_factor
        Ks[:, :2, :3] //= render_factor
    flip_x = bool(getattr(getattr(cfg, 'data', None), 'flip_x', False))
    flip_y = bool(getattr(getattr(cfg, 'data', None), 'flip_y', False))
    rgbs, disps, normals, stats = [], [], [], dict(psnr=[], psnr_fore=[], psnr_back=[])
    for i, c2w in enumerate(c2ws):
        H, W = int(HW[i][0]), int(HW[i][1])
        res = render_view(model, H, W, torch.tensor(Ks[i], device=dev), c2w, ndc, render_kwargs, flip_x, flip_y)
        rgb = res['rgb_marched'].cpu().numpy()
        rgbs.append(rgb)
        ident = idx if idx is not None else i
        pre = f'{step}_' if step > 0 else ''
        if rgb_only:
            if savedir is not None:
                write_png(os.path.join(savedir, f'{i:03d}.png'), to8b(rgb))
            continue
        disps.append(res['disp'].cpu().numpy())
        normal = res['normal_marched'].cpu().numpy() if 'normal_marched' in res else None
        normals.append(normal)
        mask = None
        if masks is not None:
            mask = masks[i].cpu().numpy() if isinstance(masks[i], torch.Tensor) else np.asarray(masks[i])
            if mask.ndim == 2:
                mask = mask.reshape(H, W, 1)
        if gt_imgs is not None and render_factor == 0:
            gt = gt_imgs[i].cpu().numpy() if isinstance(gt_imgs[i], torch.Tensor) else np.asarray(gt_imgs[i])
            p, fore, back = psnr_terms(rgb, gt, mask)
            stats['psnr'].append(p), stats['psnr_fore'].append(fore), stats['psnr_back'].append(back)
        if savedir is not None:
            os.makedirs(savedir, exist_ok=True)
            rgb8 = to8b(rgb)
            img8 = rgb8
            if gt_imgs is not None:
                err = 1 - np.exp(-20 * np.square(rgb - gt).sum(-1))[..., None].repeat(3, -1)
                gt8 = to8b(gt)
                write_png(os.path.join(savedir, f'{pre}gt_{ident:03d}.png'), gt8)
                img8 = np.concatenate([to8b(err), rgb8, gt8], axis=0)
            write_png(os.path.join(savedir, f'{pre}render_{ident:03d}.png'), rgb8)
            write_png(os.path.join(savedir, f'{pre}{ident:03d}.png'), img8)
            if normal is not None:                          # camera-frame normal map (lib/nvs_fun.py:164-173)
                rot = c2w[:3, :3].T.cpu().numpy()
                n = 0.5 - 0.5 * (rot @ normal[..., None])[..., 0]
                if mask is not None:
                    n = n * mask.mean(-1)[..., None] + (1 - mask)
                write_png(os.path.join(savedir, f'{pre}{ident:03d}_normal.png'), to8b(n))
    render_viewpoints.last = {k: (float(np.mean(v)) if v else None) for k, v in stats.items()}
    render_viewpoints.last['per_view'] = stats
    return np.array(rgbs), np.array(disps)


render_viewpoints.last = None
