import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'chainer-vq-vae_amd'), os.path.join(ROOT, 'oracle'), ROOT,
          os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def gpu():
    """Initialises the device backend; FAILS (does not skip) when the HIP
    extension or the GPU is missing -- there is no fallback to test."""
    import vqvae_amd.backend as backend
    backend.init(0)
    return backend


@pytest.fixture(params=['float32x2', 'float32x3', 'float32'])
def matmul_mode(request, gpu):
    """The three fp32-accurate matmul modes: 'float32x2' (three fp16 MFMA products per fp32 product of scaled
    two-piece operands), 'float32x3' (six bf16 MFMA products of an exact three-way split) and 'float32'
    (fp32 MFMA); 'bfloat16' has its own tests below."""
    gpu.set_matmul_dtype(request.param)
    gpu.set_f32x2_min_gflop(0.0)        # 'float32x2': the generic convs take the three-product kernels at the tests' small shapes too
    yield request.param
    gpu.set_f32x2_min_gflop(8.0)
    gpu.set_matmul_dtype(gpu.default_matmul_dtype())
