"""Round-5 fusions of egonn_forward must not change a bit: conv1 -> conv2 through split-form maps, the 1x1 downsample branch + gated
residual in one launch, the local head's lateral inside the heads' kernel — each against its measurement switch, fp32 and bf16 maps
(tools/check_bitwise_switches.py runs one process per switch: the switches are read once per process)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_fusion_switches_are_bitwise_neutral():
    import __graft_entry__ as g
    g.build()
    from egonn_amd import _lib
    _lib.require_gpu()
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "check_bitwise_switches.py")], capture_output=True, text=True,
                       timeout=900, cwd=repo)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    lines = [l for l in r.stdout.splitlines() if "bitwise equal" in l]
    assert lines == ["fp32 bitwise equal: True", "bf16 bitwise equal: True"], r.stdout[-1500:]
