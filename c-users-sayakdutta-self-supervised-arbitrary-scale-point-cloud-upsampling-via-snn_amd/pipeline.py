"""Per-cloud driver around ``Generator3D6.upsample`` — the reference's ``generate.py``.

Mirrors /root/reference/generate.py:43-101 function for function:

* ``normalize_pointcloud(cloud)``       :43-54  bounding-box centre / largest extent (host float64, a few flops)
* ``farthest_point_sample(xyz, npoint)`` :56-74  the npoint-step sampling loop — ONE persistent HIP launch here
  (csrc/fps.hip) instead of ~6 torch kernels per step
* ``process_file(input, output, generator, target_points)`` :81-101  loadtxt -> normalise -> upsample ->
  denormalise -> FPS -> savetxt("%.6f")

``process_cloud`` is the same without the two text files.  No CPU path: the HIP library is required.
"""
import numpy as np
import torch

from . import _lib


def normalize_pointcloud(cloud):
    """(cloud - bbox centre) / largest bbox extent -> (cloud, loc, scale); generate.py:43-54."""
    cloud = np.asarray(cloud)
    bbox = np.zeros((2, 3))
    bbox[0] = np.min(cloud, axis=0)
    bbox[1] = np.max(cloud, axis=0)
    loc = (bbox[0] + bbox[1]) / 2
    scale = (bbox[1] - bbox[0]).max()
    scale_inv = 1.0 / scale if scale > 0 else 1.0
    return (cloud - loc) * scale_inv, loc, scale


def farthest_point_sample_device(xyz_dev, npoint):
    """xyz_dev [N,3] float32 CUDA tensor -> int64 CUDA tensor [npoint] of sampled indices (first = N//2)."""
    if xyz_dev.dim() != 2 or xyz_dev.shape[1] != 3 or xyz_dev.dtype != torch.float32 or not xyz_dev.is_cuda:
        raise ValueError("farthest_point_sample_device: expected a CUDA float32 tensor [N,3]")
    n, npoint = xyz_dev.shape[0], int(npoint)
    if npoint < 0 or npoint > n:
        # the reference would sample duplicates past N (generate.py:93 asserts N >= target first)
        raise ValueError("farthest_point_sample: need 0 <= npoint <= N (N=%d, npoint=%d)" % (n, npoint))
    lib = _lib.load()
    xyz_dev = xyz_dev.contiguous()
    out = torch.empty((npoint,), dtype=torch.int64, device=xyz_dev.device)
    if npoint == 0:
        return out
    nbytes = int(lib.sapcu_fps_workspace_bytes(npoint))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=xyz_dev.device)
    with torch.cuda.device(xyz_dev.device):
        _lib.check(lib.sapcu_fps_f32(_lib.ptr(xyz_dev), n, npoint, _lib.ptr(out), _lib.ptr(ws), nbytes,
                                     _lib.current_stream()))
    return out


def farthest_point_sample(xyz, npoint, device="cuda"):
    """Return indices, like generate.py:56-74: xyz ndarray [N,3] -> int64 ndarray [npoint]."""
    x = torch.from_numpy(np.ascontiguousarray(xyz)).float().to(device)        # generate.py:59
    return farthest_point_sample_device(x, npoint).cpu().numpy()


def process_cloud(cloud, generator, target_points):
    """generate.py:81-99 without the files: cloud [N,>=3] -> ndarray [target_points,3] float64."""
    cloud = np.asarray(cloud)[:, :3]
    cloud, loc, scale = normalize_pointcloud(cloud)
    upsampled = np.array(generator.upsample(np.expand_dims(cloud, 0)))
    upsampled = upsampled * scale + loc
    assert upsampled.shape[0] >= target_points, \
        "Generated %d points, expected >= %d" % (upsampled.shape[0], target_points)
    indices = farthest_point_sample(upsampled, target_points, device=generator.device)
    return upsampled[indices]


def process_file(input_path, output_path, generator, target_points):
    """generate.py:81-101: text cloud in, ``target_points`` upsampled points out ("%.6f")."""
    cloud = np.loadtxt(input_path)
    np.savetxt(output_path, process_cloud(cloud, generator, target_points), fmt="%.6f")
