"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes binding + build recipe of oracle/egonn_cpu.c (C/OpenMP restatement of
the reference's per-scan descriptor extraction; see the header of that file).  Used by tests/test_oracle.py (validated
against the numpy oracle and the reference-graph fixtures) and by bench.py's `cpu_baseline` leg — never by egonn_amd/."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, List

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "egonn_cpu.c")
BUILD_DIR = os.path.join(HERE, "_build")

PLANES = [32, 64, 64, 128, 128, 128, 128]


def weight_order() -> List[str]:
    """state_dict keys in the order egonn_cpu.c consumes them (its traversal of the graph)."""
    bn = lambda p: [f"{p}.bn.weight", f"{p}.bn.bias", f"{p}.bn.running_mean", f"{p}.bn.running_var"]
    keys = ["trunk.convs.0.kernel"] + bn("trunk.bn.0")
    cin = 32
    for i, cout in enumerate(PLANES, start=1):
        b = f"trunk.blocks.{i}.0"
        keys += [f"trunk.convs.{i}.kernel"] + bn(f"trunk.bn.{i}")
        keys += [f"{b}.conv1.kernel"] + bn(f"{b}.norm1") + [f"{b}.conv2.kernel"] + bn(f"{b}.norm2")
        if cin != cout:
            keys += [f"{b}.downsample.0.kernel"] + bn(f"{b}.downsample.1")
        keys += [f"{b}.eca.conv.weight"]
        cin = cout
    mlp = lambda p: [f"{p}.net.0.linear.weight", f"{p}.net.0.linear.bias", f"{p}.net.2.linear.weight", f"{p}.net.2.linear.bias"]
    keys += ["global_head.conv1x1.7.kernel", "global_head.tconv.7.kernel", "global_head.conv1x1.6.kernel",
             "global_head.tconv.6.kernel", "global_head.conv1x1.5.kernel"]
    keys += mlp("global_descriptor_decoder") + ["global_pooling.pooling.p"]
    keys += ["local_head.conv1x1.4.kernel", "local_head.tconv.4.kernel", "local_head.conv1x1.3.kernel"]
    keys += mlp("local_descriptor_decoder") + mlp("local_keypoint_regressor") + mlp("local_sigma_regressor")
    return keys


def build(native: bool = False, force: bool = False) -> str:
    """gcc -O3 -fopenmp -shared; `native` adds -march=native (bench.py builds that flavour on the box it runs on)."""
    os.makedirs(BUILD_DIR, exist_ok=True)
    if native:      # never ships: compiled for, and kept on, the machine that runs it
        import hashlib
        import tempfile
        try:
            flags = [l for l in open("/proc/cpuinfo") if l.startswith("flags")][0]
        except Exception:
            flags = "unknown"
        out = os.path.join(tempfile.gettempdir(), f"libegonn_cpu_native_{hashlib.sha1(flags.encode()).hexdigest()[:12]}.so")
    else:
        out = os.path.join(BUILD_DIR, "libegonn_cpu.so")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(SRC):
        arch = ["-march=native"] if native else ["-mavx2", "-mfma"]
        cmd = ["gcc", "-O3", "-fopenmp", "-fPIC", "-shared", "-std=c11"] + arch + ["-o", out, SRC, "-lm"]
        subprocess.run(cmd, check=True)
    return out


class CpuOracle:
    def __init__(self, state_dict: Dict[str, np.ndarray], quant_step, native: bool = False):
        """quant_step: a float (CartesianQuantizer) or three floats (PolarQuantizer: degrees, metres, metres)."""
        self.lib = C.CDLL(build(native))
        P = C.POINTER(C.c_float)
        self.lib.egonn_cpu_compute_embedding_q.restype = C.c_int
        self.lib.egonn_cpu_compute_embedding_q.argtypes = [P, C.c_int64, C.c_int, P, C.POINTER(P), C.c_int, C.c_int, P,
                                                           C.POINTER(C.c_int32), C.POINTER(C.c_int32), P, P, P, C.c_int]
        self.lib.egonn_cpu_num_threads.restype = C.c_int
        steps = [float(v) for v in np.atleast_1d(np.asarray(quant_step, dtype=np.float64))]
        self.mode = 1 if len(steps) == 3 else 0
        self.step = (C.c_float * 3)(*(steps + steps[-1:] * 2)[:3])
        keys = weight_order()
        self._keep = [np.ascontiguousarray(np.asarray(state_dict[k], dtype=np.float32)) for k in keys]
        self._ptrs = (P * len(keys))(*[a.ctypes.data_as(P) for a in self._keep])
        self._n = len(keys)

    @property
    def threads(self) -> int:
        return int(self.lib.egonn_cpu_num_threads())

    def throughput(self, scans, workers: int, threads_per_worker: int, min_seconds: float, n_k: int = 128):
        """scans/s with `workers` scans in flight (host threads; the C call releases the GIL), each on
        `threads_per_worker` OpenMP threads; whole passes over `scans` until min_seconds."""
        import time
        from concurrent.futures import ThreadPoolExecutor
        done, t0 = 0, time.perf_counter()
        with ThreadPoolExecutor(max_workers=workers) as pool:
            while time.perf_counter() - t0 < min_seconds:
                list(pool.map(lambda sc: self.compute_embedding(sc, n_k, threads_per_worker), scans))
                done += len(scans)
        return done / (time.perf_counter() - t0), done, time.perf_counter() - t0

    def compute_embedding(self, pc: np.ndarray, n_k: int = 128, n_threads: int = 0):
        """-> (global (1,256), keypoints (m,3), descriptors (m,128), keypoint coords (m,3), sigma (m,), level counts)"""
        P = C.POINTER(C.c_float)
        pc = np.ascontiguousarray(pc, dtype=np.float32)
        g = np.empty(256, np.float32)
        cnt = np.zeros(8, np.int32)
        sc = np.zeros((n_k, 3), np.int32)
        kp = np.zeros((n_k, 3), np.float32)
        de = np.zeros((n_k, 128), np.float32)
        sg = np.zeros(n_k, np.float32)
        m = self.lib.egonn_cpu_compute_embedding_q(pc.ctypes.data_as(P), len(pc), self.mode, self.step, self._ptrs, self._n, n_k,
                                                 g.ctypes.data_as(P), cnt.ctypes.data_as(C.POINTER(C.c_int32)),
                                                 sc.ctypes.data_as(C.POINTER(C.c_int32)), kp.ctypes.data_as(P),
                                                 de.ctypes.data_as(P), sg.ctypes.data_as(P), int(n_threads))
        if m < 0:
            raise RuntimeError(f"egonn_cpu_compute_embedding failed ({m})")
        return g[None], kp[:m], de[:m], sc[:m], sg[:m], cnt
