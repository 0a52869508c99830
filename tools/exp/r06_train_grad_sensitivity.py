"""probe: gradient digests of the polar training fixture under four arithmetic settings of the sparse convolutions —
exact fp32 kernels, fp16-split unsplit offsets, the product rule (offset parts), plain one-thread kernel — each against the
fixture and against each other.  Tells conditioning (all settings scatter alike) from a defect (one setting is off)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as g; g.build()
import egonn_amd
from egonn_amd import _lib
import helpers as H
name = sys.argv[1] if len(sys.argv) > 1 else "egonn_train_polar"
case = H.load_case(name)
dev = _lib.require_gpu()
polar = str(case["coordinates"]) == "polar"
step = [float(v) for v in case["quantization_step"]]
def run(setting):
    mp = egonn_amd.ModelParams(model="egonn", coordinates="polar" if polar else "cartesian", quantization_step=step if polar else step[0])
    model = egonn_amd.model_factory(mp)
    w = H.seeded_weights(int(case["weight_seed"]))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    model = model.to(dev).train()
    ctx = model.context()
    if setting == "exact": ctx.set_exact_fp32(True)
    if setting == "naive": ctx.set_naive_conv(True)
    if setting.startswith("only"):            # product rule on ONE level (both map classes), unsplit elsewhere
        keep = int(setting[4:])
        for mc in (0, 1):
            for lv in range(8):
                if lv != keep: ctx.set_ksplit(mc, lv, kparts=1, kw=0, col_parts=0)
    if setting == "unsplit":
        for mc in (0, 1):
            for lv in range(8): ctx.set_ksplit(mc, lv, kparts=1, kw=0, col_parts=0)
    coords = torch.from_numpy(case["coords"]).to(dev)
    y = model({"coords": coords, "features": torch.ones((len(coords), 1), device=dev)})
    R = torch.from_numpy(np.random.default_rng(int(case["proj_seed"])).standard_normal(case["global"].shape).astype(np.float32)).to(dev)
    loss = (y["global"] * R).sum()
    if os.environ.get("LOCAL", "1") == "1":
        kcs = model.keypoint_coords()
        for b in range(int(case["n_scans"])):
            kc = kcs[b].cpu().numpy()
            d, k, sg = y["descriptors"][b], y["keypoints"][b], y["sigma"][b]
            def rw(width, salt):
                c = np.asarray(kc, dtype=np.float64)
                phase = 2.1 * c[:, 0] + 0.37 * c[:, 1] + 0.73 * c[:, 2] + 1.13 * c[:, 3] + salt
                return torch.from_numpy(np.cos(phase[:, None] + 0.05 * np.arange(width)[None, :]).astype(np.float32)).to(dev)
            loss = loss + (d * rw(128, 0.1)).sum() + (k * rw(3, 0.2)).sum() + (sg * rw(1, 0.3)).sum()
    loss.backward()
    return {k: p.grad.detach().double().cpu().numpy().ravel() for k, p in model.named_parameters() if p.grad is not None}
res = {s: run(s) for s in ("naive", "exact", "unsplit", "product", "only3", "only4", "only5")}
keys = ["trunk.convs.0.kernel", "trunk.convs.1.kernel", "trunk.convs.4.kernel", "trunk.blocks.5.0.conv1.kernel", "trunk.blocks.7.0.conv2.kernel", "global_head.conv1x1.7.kernel"]
for a, b in [("naive", "exact"), ("naive", "unsplit"), ("naive", "product"), ("naive", "only3"), ("naive", "only4"), ("naive", "only5")]:
    print(a, "vs", b, " ".join(f"{k.split('.')[1] if k.startswith('trunk') else 'gh'}.{k.split('.')[2]}:{np.linalg.norm(res[a][k] - res[b][k]) / max(np.linalg.norm(res[a][k]), 1e-30):.1e}" for k in keys))
