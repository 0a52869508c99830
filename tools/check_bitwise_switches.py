"""Fusions of egonn_forward that must not change a bit, each against its measurement switch (runs itself in subprocesses):
  EGONN_NO_PRESPLIT=1    conv2 splits its operands in the step loop instead of reading conv1's split-form output (sconv_split.hip)
  EGONN_NO_FUSED_DOWN=1  1x1 downsample branch + BatchNorm and the gated residual + ReLU as two launches instead of one (dense.hip)
  EGONN_NO_FUSED_LATERAL=1  the local head's level-3 lateral 1x1 convolution as its own launch instead of the heads' first layer
  EGONN_NO_GATED_K2S2=1  level 1's block tail as its own launch + a 23 MB map instead of being evaluated by level 2's strided convolution
  EGONN_NO_FUSED_GHEAD=1  MinkHead's three lateral 1x1 convolutions as three launches and every FPN add inside the next lateral (round 5) instead of
                          one grouped launch + the lateral as the transposed convolution's epilogue residual (fp32 maps)
fp32 maps (4 scans) and bf16 maps."""
import os, subprocess, sys, hashlib
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as g
    g.build()
    import egonn_amd as E
    from egonn_amd.synth import lidar_scan, seeded_state_dict
    mp = E.ModelParams(model="egonn", coordinates="cartesian", quantization_step=0.1)
    m = E.model_factory(mp)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(7, shapes).items()})
    m = m.to("cuda").eval()
    m.coord_bits = 12
    m.precision = os.environ.get("CHECK_PRECISION", "fp32")
    scans = [lidar_scan(300 + i, 50000) for i in range(4)]
    off = [0]
    for s in scans:
        off.append(off[-1] + len(s))
    pts = torch.from_numpy(np.concatenate(scans)).cuda()
    out = E.DescriptorExtractor(m, n_k=128).extract_packed(pts, off)
    h = hashlib.sha256()
    for k in ("global", "keypoints", "descriptors", "rows"):
        h.update(out[k].cpu().numpy().tobytes())
    for t in m._last_local:                            # every row of the three local heads
        h.update(t.cpu().numpy().tobytes())
    for l in (2, 3, 4, 5):                              # (level 1's block output is not materialised in the product path)
        ctx = m.context(0)
        h.update(ctx.forward_level_features(l, [0, 32, 64, 64, 128, 128][l]).cpu().numpy().tobytes())
        torch.cuda.synchronize()
    print("DIGEST", h.hexdigest())
else:
    ok = True
    for prec in ("fp32", "bf16"):
        d = []
        for env in ({}, {"EGONN_NO_PRESPLIT": "1"}, {"EGONN_NO_FUSED_DOWN": "1"}, {"EGONN_NO_FUSED_LATERAL": "1"}, {"EGONN_NO_GATED_K2S2": "1"}, {"EGONN_NO_FUSED_GHEAD": "1"}):
            r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True,
                               env=dict(os.environ, CHECK_PRECISION=prec, **env))
            line = [l for l in r.stdout.splitlines() if l.startswith("DIGEST")]
            assert line, r.stderr[-2000:]
            d.append(line[0])
            print(prec, env, line[0])
        same = len(set(d)) == 1
        print(prec, "bitwise equal:", same)
        ok &= same
    sys.exit(0 if ok else 1)
