"""conv1 -> conv2 through the split-form map (sconv_split.hip, egonn_forward) against the in-loop split (EGONN_NO_PRESPLIT=1):
the two must be BITWISE equal (the same split8h makes the hi / lo parts either way).  Runs itself twice in subprocesses."""
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
    scans = [lidar_scan(300 + i, 50000) for i in range(4)]
    off = [0]
    for s in scans:
        off.append(off[-1] + len(s))
    pts = torch.from_numpy(np.concatenate(scans)).cuda()
    out = E.DescriptorExtractor(m, n_k=128).extract_packed(pts, off)
    h = hashlib.sha256()
    for k in ("global", "keypoints", "descriptors", "rows"):
        h.update(out[k].cpu().numpy().tobytes())
    for l in (1, 2, 3, 4, 5):
        ctx = m.context(0)
        h.update(ctx.forward_level_features(l, [0, 32, 64, 64, 128, 128][l]).cpu().numpy().tobytes())
    print("DIGEST", h.hexdigest())
else:
    d = []
    for env in ({}, {"EGONN_NO_PRESPLIT": "1"}):
        r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, env=dict(os.environ, **env))
        line = [l for l in r.stdout.splitlines() if l.startswith("DIGEST")]
        assert line, r.stderr[-2000:]
        d.append(line[0])
        print(env, line[0])
    print("bitwise equal:", d[0] == d[1])
    sys.exit(0 if d[0] == d[1] else 1)
