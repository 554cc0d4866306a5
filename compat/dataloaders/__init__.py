"""Stand-in for the DataLoaders package (`from dataloaders import *`, train_permuto_sdf.py:19; common_utils.py:410-500).

`DataLoaderDTU` has the setter API the reference drives (common_utils.py:463-499) and two sources of frames:

* an on-disk scene in the NeuS layout the reference's data download uses (``<dataset_path>/<scene>/cameras_sphere.npz`` +
  ``image/*.png`` [+ ``mask/*.png``]): projection matrices ``world_mat_i @ scale_mat_i`` are decomposed into K and the
  camera pose (RQ decomposition), the scene is rotated about x and scaled exactly as ``loader_dtu`` of
  config/train_permuto_sdf.cfg asks (``rotate_scene_x_axis_degrees``, ``scene_scale_multiplier``);
* when that path does not exist (no dataset offline): a SYNTHETIC scene of the same shape -- 49 pin-hole cameras on a
  sphere looking at the origin, images ray-traced from an analytic scene (a radius-0.3 shaded, textured sphere in front of
  a smooth direction-dependent background).  Resolution: env ``PSDF_SYNTH_RES=WxH`` (default 400x300; DTU is 1600x1200).

``MiscDataFuncs.frames2tensors`` stacks frames into the TensorReel that ``PermutoSDF.random_rays_from_reel`` reads
(src/PermutoSDF.cu:70-102; PermutoSDFGPU.cuh:65-85: K row-major, tf_world_cam row-major with [R|t] in rows 0-2).
The other loaders of the package raise when constructed."""
import math
import os
import re

import numpy as np


def _loader(name):
    def __init__(self, *a, **k):
        raise RuntimeError("dataloaders stand-in: %s needs the DataLoaders package and its dataset (compat/README.md)" % name)
    return type(name, (), {"__init__": __init__})


for _n in ["DataLoaderEasyPBR", "DataLoaderMultiFace", "DataLoaderPhenorobCP1", "DataLoaderNerf",
           "DataLoaderColmap", "DataLoaderBlenderFB", "DataLoaderShapeNetImg", "DataLoaderSRN", "DataLoaderLLFF"]:
    globals()[_n] = _loader(_n)


# ------------------------------------------------------------------------------------------------ small math types
class Affine3:
    """Rigid/affine transform with the few Eigen::Affine3 members the reference's Python touches
    (nerf_utils.py:488-490: inverse(), linear(), translation(); matrix())."""

    def __init__(self, m=None):
        self.m = np.eye(4, dtype=np.float64) if m is None else np.asarray(m, dtype=np.float64).reshape(4, 4).copy()

    def matrix(self):
        return self.m.astype(np.float32)

    def inverse(self):
        return Affine3(np.linalg.inv(self.m))

    def linear(self):
        return self.m[:3, :3].astype(np.float32)

    def translation(self):
        return self.m[:3, 3].astype(np.float32)

    def clone(self):
        return Affine3(self.m)

    def __matmul__(self, o):
        return Affine3(self.m @ o.m)


class Frame:
    """What the reference reads from an easypbr Frame: width, height, K [3,3] float32, tf_cam_world, frame_idx, cam_id,
    rgb_32f / mask as float32 HxWxC arrays, is_shell, subsample(), load_images()."""

    def __init__(self):
        self.width = self.height = 0
        self.K = np.eye(3, dtype=np.float32)
        self.tf_cam_world = Affine3()
        self.frame_idx = self.cam_id = 0
        self.rgb_32f = None          # [H, W, 3] float32 in [0, 1]
        self.mask = None             # [H, W, 1] float32 or None
        self.is_shell = False
        self.rgb_path = ""

    def load_images(self):
        self.is_shell = False

    def has_extra_field(self, name):
        return False

    def pos_in_world(self):
        return self.tf_cam_world.inverse().translation()

    def look_dir(self):
        return self.tf_cam_world.inverse().linear()[:, 2]

    def subsample(self, factor, subsample_imgs=True):
        f = Frame()
        f.__dict__.update(self.__dict__)
        f.width, f.height = int(self.width / factor), int(self.height / factor)
        K = self.K.astype(np.float64).copy()
        K[:2, :] /= factor
        f.K = K.astype(np.float32)
        if subsample_imgs and self.rgb_32f is not None:
            s = int(round(factor))
            f.rgb_32f = self.rgb_32f[::s, ::s][:f.height, :f.width].copy()
            f.mask = None if self.mask is None else self.mask[::s, ::s][:f.height, :f.width].copy()
        return f

    def project(self, p):
        pc = self.tf_cam_world.m @ np.append(np.asarray(p, dtype=np.float64).reshape(3), 1.0)
        uv = self.K.astype(np.float64) @ pc[:3]
        return (uv / uv[2]).astype(np.float32)


class TensorReel:
    def __init__(self, rgb_reel=None, mask_reel=None, K_reel=None, tf_world_cam_reel=None, has_mask=False,
                 tf_cam_world_reel=None):
        self.rgb_reel, self.mask_reel, self.K_reel, self.tf_world_cam_reel = rgb_reel, mask_reel, K_reel, tf_world_cam_reel
        self.tf_cam_world_reel = tf_cam_world_reel
        self.has_mask = has_mask


class MiscDataFuncs:
    @staticmethod
    def frames2tensors(frames):
        """-> TensorReel on the current CUDA device (the reference's kernels read it there)."""
        import torch
        if len(frames) == 0:
            raise RuntimeError("frames2tensors: no frames")
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        has_mask = all(f.mask is not None for f in frames)
        rgb = torch.stack([torch.as_tensor(np.ascontiguousarray(f.rgb_32f)).permute(2, 0, 1) for f in frames]).float()
        if has_mask:
            mask = torch.stack([torch.as_tensor(np.ascontiguousarray(f.mask)).view(f.height, f.width, 1).permute(2, 0, 1)
                                for f in frames]).float()
        else:
            mask = torch.ones((len(frames), 1, frames[0].height, frames[0].width))
        K = torch.stack([torch.as_tensor(np.asarray(f.K, dtype=np.float32)) for f in frames])
        tcw = torch.stack([torch.as_tensor(f.tf_cam_world.matrix()) for f in frames])
        twc = torch.stack([torch.as_tensor(f.tf_cam_world.inverse().matrix()) for f in frames])
        return TensorReel(rgb.contiguous().to(dev), mask.contiguous().to(dev), K.contiguous().to(dev),
                          twc.contiguous().to(dev), has_mask, tcw.contiguous().to(dev))


# ------------------------------------------------------------------------------------------------ config block
def _cfg_block(config_path, block):
    """`block: { key: value ... }` of the reference's configuru files as a flat dict of strings (comments stripped)."""
    try:
        txt = open(config_path).read()
    except (OSError, TypeError):
        return {}
    txt = re.sub(r"//[^\n]*", "", txt)
    m = re.search(block + r"\s*:\s*\{", txt)
    if not m:
        return {}
    depth, i = 1, m.end()
    while i < len(txt) and depth:
        depth += {"{": 1, "}": -1}.get(txt[i], 0)
        i += 1
    out = {}
    for k, v in re.findall(r"^\s*([A-Za-z_0-9]+)\s*:\s*([^\n{]+?)\s*$", txt[m.end():i - 1], re.M):
        out.setdefault(k, v.strip().strip('"'))
    return out


# ------------------------------------------------------------------------------------------------ analytic scene
SYNTH_RADIUS = 0.3
_LIGHT = np.array([0.4, 0.8, 0.45]) / np.linalg.norm([0.4, 0.8, 0.45])


def synthetic_radiance(origins, dirs):
    """Colour seen along rays [N,3] (numpy float64): the radius-0.3 sphere (albedo pattern x lambert + ambient) over a
    smooth background.  Returns rgb [N,3] in [0,1] and hit mask [N,1]."""
    o, d = np.asarray(origins, np.float64), np.asarray(dirs, np.float64)
    b = (o * d).sum(1)
    c = (o * o).sum(1) - SYNTH_RADIUS ** 2
    disc = b * b - c
    hit = (disc > 0) & (-b - np.sqrt(np.maximum(disc, 0)) > 0)
    t = -b - np.sqrt(np.maximum(disc, 0))
    p = o + d * t[:, None]
    n = p / SYNTH_RADIUS
    albedo = 0.5 + 0.5 * np.stack([np.sin(9 * n[:, 0] + 1.0), np.sin(7 * n[:, 1] + 2.0), np.sin(11 * n[:, 2])], 1) * 0.6
    shade = 0.35 + 0.65 * np.clip(n @ _LIGHT, 0, 1)
    fg = np.clip(albedo * shade[:, None], 0, 1)
    bg = 0.5 + 0.35 * np.stack([d[:, 0], d[:, 1], d[:, 2]], 1)
    rgb = np.where(hit[:, None], fg, bg)
    return rgb.astype(np.float32), hit[:, None].astype(np.float32)


def _look_at_cam_world(eye, up=(0.0, 1.0, 0.0)):
    """tf_cam_world (4x4) of a camera at `eye` looking at the origin; camera axes x right, y down, z forward."""
    eye = np.asarray(eye, np.float64)
    z = -eye / np.linalg.norm(eye)
    x = np.cross(z, np.asarray(up, np.float64))
    if np.linalg.norm(x) < 1e-6:
        x = np.cross(z, np.array([1.0, 0.0, 0.0]))
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    twc = np.eye(4)
    twc[:3, 0], twc[:3, 1], twc[:3, 2], twc[:3, 3] = x, y, z, eye
    return np.linalg.inv(twc)


def synthetic_frames(nr_images=49, width=400, height=300, cam_radius=1.3, seed=0, with_mask=False):
    rng = np.random.default_rng(seed)
    frames = []
    f = 1.1 * width  # ~50 degree horizontal field of view: the unit-sphere scene fills the image like a DTU object
    ys, xs = np.meshgrid(np.arange(height) + 0.5, np.arange(width) + 0.5, indexing="ij")
    for i in range(nr_images):
        # spiral over the upper hemisphere + a little jitter, like a DTU camera arc
        u = (i + 0.5) / nr_images
        theta, phi = math.acos(1 - 1.2 * u), 2 * math.pi * i * 0.61803398875
        eye = cam_radius * np.array([math.sin(theta) * math.cos(phi), math.cos(theta), math.sin(theta) * math.sin(phi)])
        eye += rng.normal(scale=0.02, size=3)
        fr = Frame()
        fr.width, fr.height, fr.frame_idx, fr.cam_id = width, height, i, i
        fr.K = np.array([[f, 0, width / 2], [0, f, height / 2], [0, 0, 1]], np.float32)
        fr.tf_cam_world = Affine3(_look_at_cam_world(eye))
        twc = np.linalg.inv(fr.tf_cam_world.m)
        pc = np.stack([(xs - width / 2) / f, (ys - height / 2) / f, np.ones_like(xs)], -1).reshape(-1, 3)
        d = pc @ twc[:3, :3].T
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rgb, hit = synthetic_radiance(np.broadcast_to(twc[:3, 3], d.shape), d)
        fr.rgb_32f = rgb.reshape(height, width, 3)
        fr.mask = hit.reshape(height, width, 1) if with_mask else None
        frames.append(fr)
    return frames


# ------------------------------------------------------------------------------------------------ NeuS-layout scene
def _decompose_projection(P):
    """3x4 projection -> (K 3x3 with K[2,2]=1, tf_world_cam 4x4) via RQ of the left 3x3 (what cv2.decomposeProjectionMatrix
    does in the reference's loader)."""
    from scipy.linalg import rq
    M = P[:3, :3]
    K, R = rq(M)
    S = np.diag(np.sign(np.diag(K)))
    K, R = K @ S, S @ R
    if np.linalg.det(R) < 0:
        R, K = -R, -K
    c = -np.linalg.inv(M) @ P[:3, 3]
    K = K / K[2, 2]
    twc = np.eye(4)
    twc[:3, :3], twc[:3, 3] = R.T, c
    return K, twc


def neus_scene_frames(scene_dir, load_mask, subsample, scale_mult, rot_x_deg):
    from PIL import Image
    cams = np.load(os.path.join(scene_dir, "cameras_sphere.npz"))
    names = sorted(n for n in os.listdir(os.path.join(scene_dir, "image")) if n.lower().endswith((".png", ".jpg")))
    a = math.radians(rot_x_deg)
    rot = np.eye(4)
    rot[1, 1], rot[1, 2], rot[2, 1], rot[2, 2] = math.cos(a), -math.sin(a), math.sin(a), math.cos(a)
    frames = []
    for i, name in enumerate(names):
        P = (cams["world_mat_%d" % i] @ cams["scale_mat_%d" % i])[:3, :4]
        K, twc = _decompose_projection(P)
        twc = rot @ twc                    # rotate the (unit-sphere normalised) scene about x ...
        twc[:3, 3] *= scale_mult           # ... and shrink it into the radius-0.5 bounding sphere
        img = np.asarray(Image.open(os.path.join(scene_dir, "image", name)).convert("RGB"), np.float32) / 255.0
        fr = Frame()
        fr.height, fr.width = img.shape[:2]
        fr.K, fr.tf_cam_world, fr.frame_idx, fr.cam_id = K.astype(np.float32), Affine3(np.linalg.inv(twc)), i, i
        fr.rgb_32f, fr.rgb_path = img, os.path.join(scene_dir, "image", name)
        mp = os.path.join(scene_dir, "mask", name)
        if load_mask and os.path.exists(mp):
            fr.mask = (np.asarray(Image.open(mp).convert("L"), np.float32) / 255.0)[:, :, None]
        if subsample > 1:
            fr = fr.subsample(subsample, True)
        frames.append(fr)
    return frames


class DataLoaderDTU:
    """Setter API of common_utils.py:463-499; `start()` loads (or synthesises) every frame, there is no reader thread."""

    def __init__(self, config_path=None):
        cfg = _cfg_block(config_path, "loader_dtu")
        self.m_dataset_path = cfg.get("dataset_path", "")
        self.m_scene = cfg.get("restrict_to_scene_name", "")
        self.m_mode = cfg.get("mode", "all")
        self.m_load_mask = cfg.get("load_mask", "false") == "true"
        self.m_subsample = float(cfg.get("subsample_factor", 1))
        self.m_scale_mult = float(cfg.get("scene_scale_multiplier", 0.4))
        self.m_rot_x = float(cfg.get("rotate_scene_x_axis_degrees", 115))
        self.m_frames = []
        self.m_idx = 0
        self.is_synthetic = False

    # -- setters (common_utils.py:463-481)
    def set_dataset_path(self, p): self.m_dataset_path = p
    def set_restrict_to_scene_name(self, s): self.m_scene = s
    def set_subsample_factor(self, f): self.m_subsample = float(f)
    def set_mode_train(self): self.m_mode = "train"
    def set_mode_test(self): self.m_mode = "test"
    def set_mode_validation(self): self.m_mode = "val"
    def set_mode_all(self): self.m_mode = "all"
    def set_load_mask(self, v): self.m_load_mask = bool(v)
    def get_restrict_to_scene_name(self): return self.m_scene

    def start(self):
        scene_dir = os.path.join(self.m_dataset_path or "", self.m_scene or "")
        if os.path.exists(os.path.join(scene_dir, "cameras_sphere.npz")):
            frames = neus_scene_frames(scene_dir, self.m_load_mask, int(self.m_subsample), self.m_scale_mult, self.m_rot_x)
        else:
            w, h = (int(v) for v in os.environ.get("PSDF_SYNTH_RES", "400x300").lower().split("x"))
            s = max(1, int(self.m_subsample))
            frames = synthetic_frames(int(os.environ.get("PSDF_SYNTH_IMAGES", 49)), w // s, h // s, with_mask=self.m_load_mask)
            self.is_synthetic = True
        # every 8th image is held out for testing, like NeuS-style splits
        if self.m_mode == "train":
            frames = [f for i, f in enumerate(frames) if i % 8 != 7] if len(frames) > 8 else frames
        elif self.m_mode in ("test", "val"):
            frames = [f for i, f in enumerate(frames) if i % 8 == 7] or frames[-1:]
        self.m_frames = frames

    # -- readers
    def finished_reading_scene(self): return True
    def has_data(self): return self.m_idx < len(self.m_frames)
    def is_finished(self): return self.m_idx >= len(self.m_frames)
    def reset(self): self.m_idx = 0
    def nr_samples(self): return len(self.m_frames)
    def get_all_frames(self): return list(self.m_frames)
    def get_frame_at_idx(self, i): return self.m_frames[i]
    def get_random_frame(self): return self.m_frames[np.random.randint(len(self.m_frames))]
    def get_closest_frame(self, frame): return self.m_frames[0]

    def get_next_frame(self):
        f = self.m_frames[self.m_idx]
        self.m_idx += 1
        return f


__all__ = [n for n in list(globals()) if n.startswith("DataLoader")] + ["TensorReel", "MiscDataFuncs", "Frame", "Affine3"]
