"""compat/dataloaders: the DataLoaderDTU stand-in (setter API of common_utils.py:463-499), the synthetic scene, the
NeuS-layout on-disk loader and frames2tensors -> TensorReel (fields of src/PermutoSDF.cu:70-102).  CPU only."""
import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dl():
    spec = importlib.util.spec_from_file_location("psdf_compat_dataloaders", os.path.join(ROOT, "compat", "dataloaders", "__init__.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_synthetic_loader_follows_the_reference_setter_api(dl, monkeypatch):
    monkeypatch.setenv("PSDF_SYNTH_RES", "64x48")
    monkeypatch.setenv("PSDF_SYNTH_IMAGES", "9")
    lo = dl.DataLoaderDTU("/nonexistent.cfg")
    lo.set_dataset_path("/media/rosu/Data/data/permuto_sdf_data/data_DTU")   # comp_1 of paths/data_paths.py: absent offline
    lo.set_mode_train()
    lo.set_load_mask(False)
    lo.set_restrict_to_scene_name("dtu_scan24")
    lo.start()
    assert lo.is_synthetic and lo.nr_samples() == 8          # every 8th image is held out
    fr = lo.get_all_frames()[0]
    assert (fr.width, fr.height) == (64, 48) and fr.rgb_32f.shape == (48, 64, 3) and fr.rgb_32f.dtype == np.float32
    assert fr.K.shape == (3, 3) and fr.K.dtype == np.float32
    # camera looks at the origin: the origin projects to the principal point
    uv = fr.project([0, 0, 0])
    assert abs(uv[0] - 32) < 1e-3 and abs(uv[1] - 24) < 1e-3
    # the centre pixel sees the sphere (a hit: not the smooth background colour of that direction)
    twc = fr.tf_cam_world.inverse()
    rgb, hit = dl.synthetic_radiance(twc.translation()[None], twc.linear()[:, 2][None])
    assert hit[0, 0] == 1.0
    half = fr.subsample(2.0, subsample_imgs=False)
    assert (half.width, half.height) == (32, 24) and abs(half.K[0, 0] - fr.K[0, 0] / 2) < 1e-4


def test_frames2tensors_layout(dl, monkeypatch):
    import torch
    frames = dl.synthetic_frames(3, 32, 24)
    reel = dl.MiscDataFuncs.frames2tensors(frames)
    assert tuple(reel.rgb_reel.shape) == (3, 3, 24, 32) and tuple(reel.mask_reel.shape) == (3, 1, 24, 32)
    assert tuple(reel.K_reel.shape) == (3, 3, 3) and tuple(reel.tf_world_cam_reel.shape) == (3, 4, 4)
    assert not reel.has_mask
    # tf_world_cam: rows 0-2 are [R|t], t = camera centre at distance ~1.3, third column of R points to the origin
    twc = reel.tf_world_cam_reel[0].cpu().numpy()
    c, z = twc[:3, 3], twc[:3, 2]
    assert abs(np.linalg.norm(c) - 1.3) < 0.1 and np.dot(z, -c / np.linalg.norm(c)) > 0.999
    prod = reel.tf_world_cam_reel[0].cpu() @ reel.tf_cam_world_reel[0].cpu()
    assert torch.allclose(prod, torch.eye(4), atol=1e-5)
    assert torch.equal(reel.rgb_reel[1, :, 5, 7].cpu(), torch.as_tensor(frames[1].rgb_32f[5, 7]))


def test_neus_layout_scene_round_trip(dl, tmp_path):
    """write a 2-image scene in the NeuS layout (cameras_sphere.npz + image/*.png) from known cameras, load it back"""
    from PIL import Image
    scene = tmp_path / "data_DTU" / "dtu_scanX"
    (scene / "image").mkdir(parents=True)
    (scene / "mask").mkdir()
    frames = dl.synthetic_frames(2, 40, 30, with_mask=True)
    cams = {}
    for i, f in enumerate(frames):
        Image.fromarray((f.rgb_32f * 255 + 0.5).astype(np.uint8)).save(scene / "image" / ("%03d.png" % i))
        Image.fromarray((f.mask[:, :, 0] * 255).astype(np.uint8)).save(scene / "mask" / ("%03d.png" % i))
        P = np.eye(4)
        P[:3, :4] = f.K.astype(np.float64) @ f.tf_cam_world.m[:3, :4]
        cams["world_mat_%d" % i] = P
        cams["scale_mat_%d" % i] = np.eye(4)
    np.savez(scene / "cameras_sphere.npz", **cams)
    lo = dl.DataLoaderDTU(None)
    lo.set_dataset_path(str(tmp_path / "data_DTU"))
    lo.set_restrict_to_scene_name("dtu_scanX")
    lo.set_load_mask(True)
    lo.m_rot_x, lo.m_scale_mult = 0.0, 1.0        # identity scene transform: cameras must come back as written
    lo.start()
    assert not lo.is_synthetic and lo.nr_samples() == 2
    for f0, f1 in zip(frames, lo.get_all_frames()):
        assert np.allclose(f1.K, f0.K, atol=1e-3)
        assert np.allclose(f1.tf_cam_world.m, f0.tf_cam_world.m, atol=1e-5)
        assert np.abs(f1.rgb_32f - f0.rgb_32f).max() <= 1.0 / 255 + 1e-6
        assert f1.mask.shape == (30, 40, 1)
    # the configured scene transform: rotation about x by 115 degrees and scale 0.4 (config/train_permuto_sdf.cfg loader_dtu)
    lo2 = dl.DataLoaderDTU(None)
    lo2.set_dataset_path(str(tmp_path / "data_DTU"))
    lo2.set_restrict_to_scene_name("dtu_scanX")
    lo2.start()
    c0 = np.linalg.inv(frames[0].tf_cam_world.m)[:3, 3]
    c1 = lo2.get_all_frames()[0].tf_cam_world.inverse().translation()
    assert abs(np.linalg.norm(c1) - 0.4 * np.linalg.norm(c0)) < 1e-4


def test_mesh_export_marching_tetrahedra_and_ply(tmp_path):
    """compat/skimage.measure.marching_cubes + compat/easypbr.Mesh.save_to_file: the pieces the reference's mesh export
    (sdf_utils.py:252-292, create_my_meshes.py:162) needs; on an analytic sphere: watertight, genus 0, right area."""
    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "compat", *rel))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    measure = load("psdf_compat_measure", ("skimage", "measure.py"))
    n = 64
    t = np.linspace(-0.5, 0.5, n, dtype=np.float32)
    x, y, z = np.meshgrid(t, t, t, indexing="ij")
    vol = np.sqrt(x * x + y * y + z * z) - 0.3
    V, F, N, vals = measure.marching_cubes(vol, 0.0)
    assert V.dtype == np.float32 and F.dtype == np.int32 and V.shape[1] == 3 and F.shape[1] == 3 and len(N) == len(V)
    P = V / (n - 1.0) - 0.5                                        # the reference's index -> world mapping (sdf_utils.py:279)
    assert np.abs(np.linalg.norm(P, axis=1) - 0.3).max() < 1e-3
    e = np.sort(np.concatenate([F[:, [0, 1]], F[:, [1, 2]], F[:, [2, 0]]]), 1)
    u, c = np.unique(e, axis=0, return_counts=True)
    assert (c == 2).all() and len(V) - len(u) + len(F) == 2        # closed 2-manifold, Euler characteristic of a sphere
    fn = np.cross(P[F[:, 1]] - P[F[:, 0]], P[F[:, 2]] - P[F[:, 0]])
    assert ((fn * P[F].mean(1)).sum(1) > 0).all()                  # faces wind outwards
    assert ((N * P).sum(1) < 0).all()                              # 'descent' normals (the reference negates them)
    assert abs(0.5 * np.linalg.norm(fn, axis=1).sum() - 4 * np.pi * 0.09) < 2e-3
    with pytest.raises(ValueError):
        measure.marching_cubes(vol, 10.0)
    sys.path.insert(0, os.path.join(ROOT, "compat"))
    try:
        easypbr = load("psdf_compat_easypbr", ("easypbr", "__init__.py"))
    finally:
        sys.path.remove(os.path.join(ROOT, "compat"))
    mesh = easypbr.Mesh()
    mesh.V, mesh.F, mesh.NV = P, F, -N
    out = tmp_path / "sphere.ply"
    mesh.save_to_file(str(out))
    raw = out.read_bytes()
    head, body = raw.split(b"end_header\n")
    assert b"element vertex %d" % len(V) in head and b"element face %d" % len(F) in head and b"property float nx" in head
    assert len(body) == len(V) * 24 + len(F) * 13
    back = np.frombuffer(body[:len(V) * 24], dtype="<f4").reshape(-1, 6)
    assert np.array_equal(back[:, :3], P.astype(np.float32))
    mesh.save_to_file(str(tmp_path / "sphere.obj"))
    assert (tmp_path / "sphere.obj").read_text().count("\nf ") + 1 >= len(F)
