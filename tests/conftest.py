import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

FN_KW = dict(k_values=[24, 18, 12], emb_dims=640, time_steps_enc=4, time_steps_dec=12, num_heads=8,
             use_snn_decoder=False, decoder_dropout=0.1)
FD_KW = dict(k=32, emb_dims=768, time_steps_enc=4, time_steps_dec=8, num_heads=8, dropout=0.1,
             use_snn_decoder=False, k_scales=[8, 16, 32, 48])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def weights():
    """Conditioned state dicts (cpu tensors) for the default fn / fd configs, rebuilt from seed 0 +
    the committed BatchNorm calibration."""
    import sapcu_amd
    from sapcu_amd import testing as T

    def make(kind, **over):
        if kind == "fn":
            m = sapcu_amd.ImprovedSNNNormalEstimation(**dict(FN_KW, **over))
            bn = dict(golden("bn_calib_fn.npz"))
        else:
            m = sapcu_amd.EnhancedSNNDistanceEstimation(**dict(FD_KW, **over))
            bn = dict(golden("bn_calib_fd.npz"))
        return T.conditioned_state_dict(m.state_dict(), 0, bn_stats=bn)

    return make
