"""sapcu_amd — MI355X-native hot path of the SNN point-cloud upsampler.

Public surface = the reference's own interface for this path:
  ImprovedSNNNormalEstimation, EnhancedSNNDistanceEstimation  (fn/fd ``snn_coder``)
  Generator3D6, SNNPointCloudGenerator                        (``generation``)
  normalize_pointcloud, farthest_point_sample, process_file   (``generate``)
Importing the package never touches the GPU; the HIP library is loaded on first use and its
absence raises ``SapcuLibraryError`` (no CPU fallback exists).
"""
from ._lib import SapcuError, SapcuLibraryError, LIB_PATH  # noqa: F401
from .modules import ImprovedSNNNormalEstimation, EnhancedSNNDistanceEstimation  # noqa: F401
from .generation import Generator3D6, SNNPointCloudGenerator  # noqa: F401
from .pipeline import normalize_pointcloud, farthest_point_sample, process_cloud, process_file  # noqa: F401

__all__ = ["ImprovedSNNNormalEstimation", "EnhancedSNNDistanceEstimation", "Generator3D6",
           "SNNPointCloudGenerator", "SapcuError", "SapcuLibraryError"]
