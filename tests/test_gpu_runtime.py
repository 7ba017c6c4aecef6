"""One HIP runtime per process: the library and PyTorch must be usable in either load order (PyTorch wheels
bundle their own libamdhip64; two runtimes in one process do not both see the GPU)."""
import subprocess
import sys

import pytest

from conftest import PKG_DIR

pytestmark = pytest.mark.gpu

_SNIPPETS = {
    "library_first": "import ctypes, _native\n"
                     "n = ctypes.c_int(0); _native.check(_native.lib().stereo_hip_device_count(ctypes.byref(n)))\n"
                     "assert n.value >= 1\n"
                     "import torch\n"
                     "assert torch.zeros(4, device='cuda').sum().item() == 0.0\n",
    "torch_first": "import torch\n"
                   "assert torch.zeros(4, device='cuda').sum().item() == 0.0\n"
                   "import ctypes, _native\n"
                   "n = ctypes.c_int(0); _native.check(_native.lib().stereo_hip_device_count(ctypes.byref(n)))\n"
                   "assert n.value >= 1\n",
}


@pytest.mark.parametrize("order", sorted(_SNIPPETS))
def test_library_and_torch_share_one_runtime(order):
    code = f"import sys; sys.path.insert(0, {PKG_DIR!r})\n" + _SNIPPETS[order] + (
        "maps = {l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}\n"
        "assert len(maps) == 1, maps\nprint('ok')\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
