"""One HIP runtime per process: PyTorch's wheel brings its own copy, the library asks for the system's.  Whichever import
order a program uses, both must see the GPU (raytracer_amd.capi loads torch's copy first where torch is installed)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROGRAMS = {
    "library first": "from raytracer_amd.renderer import Context\nc = Context(0)\nimport torch\n"
                     "x = torch.ones(8, device='cuda')\nprint('ok', float(x.sum()))\nc.close()\n",
    "torch first": "import torch\nx = torch.ones(8, device='cuda')\nfrom raytracer_amd.renderer import Context\nc = Context(0)\n"
                   "print('ok', float(x.sum()))\nc.close()\n",
}


@pytest.mark.parametrize("order", sorted(PROGRAMS))
def test_library_and_torch_share_the_gpu_in_either_import_order(order):
    r = subprocess.run([sys.executable, "-c", PROGRAMS[order]], cwd=ROOT, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0 and "ok 8.0" in r.stdout, (r.stdout + r.stderr)[-2000:]
