"""Generates the golden fixtures in this directory.  RUNS ONLY IN THE BUILD CONTAINER
(needs /root/reference); the fixtures it writes are plain data and are committed.

What it does (SURVEY.md §8c "Oracle plan", Appendix D recipe):
  * puts oracle/ on sys.path so that `import MinkowskiEngine` resolves to the build's CPU
    stand-in (oracle/MinkowskiEngine — NOT the real ME, which is absent from the image);
  * shadows the HuggingFace `datasets` package with the reference's namespace package;
  * imports the REFERENCE's own misc.utils.ModelParams / models.model_factory /
    datasets.quantization and runs its graph code on seeded clouds + seeded weights;
  * stores inputs, outputs and the state_dict key/shape table as .npz / .json.

No reference source text is stored — only arrays the reference code computed.

    python tests/golden/make_golden.py            # rewrite the inference fixtures
    python tests/golden/make_golden.py train      # rewrite the train-step fixtures
    python tests/golden/make_golden.py only NAME  # rewrite one inference fixture
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def bootstrap_reference():
    assert os.path.isdir(REF), "fixture generation needs /root/reference (build container only)"
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    sys.path.insert(0, REF)
    sys.path.insert(0, REPO)
    for name, path in [("datasets", "datasets"), ("datasets.kitti", "datasets/kitti"),
                       ("datasets.mulran", "datasets/mulran"), ("datasets.southbay", "datasets/southbay")]:
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, path)]
        sys.modules[name] = m


def model_params(coordinates: str, step):
    from misc.utils import ModelParams
    f = tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False)
    f.write(f"[MODEL]\nmodel = egonn\ncoordinates = {coordinates}\nquantization_step = {step}\n")
    f.close()
    mp = ModelParams(f.name)
    os.unlink(f.name)
    return mp


def minkloc_params(model: str, step: str, block: str = "BasicBlock"):
    from misc.utils import ModelParams
    f = tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False)
    f.write(f"[MODEL]\nmodel = {model}\ncoordinates = cartesian\nquantization_step = {step}\n"
            f"planes = 32,64,64\nlayers = 1,1,1\nnum_top_down = 1\nconv0_kernel_size = 5\nfeature_size = 256\n"
            f"block = {block}\npooling = GeM\n")
    f.close()
    mp = ModelParams(f.name)
    os.unlink(f.name)
    return mp


MINKLOC_CASES = [
    # name,                 model,       block,            step,  scans,                      weight seed
    ("minkloc3d_cart03_b2", "MinkLoc3D", "BasicBlock",     "0.3", [(7, 30000), (8, 20000)],  21),
    ("minkloc_eca_cart03",  "MinkLoc",   "ECABasicBlock",  "0.3", [(9, 30000)],              22),
]


def kitti_like_filter(pc):
    """drop all-zero points, keep z > -1.5 (reference datasets/kitti/kitti_raw.py:12-14,
    misc/point_clouds.py:103-109) — applied to the synthetic cloud of the C0 case."""
    import numpy as np
    mask = ~np.all(pc == 0, axis=1)
    pc = pc[mask]
    return pc[pc[:, 2] > -1.5]


CASES = [
    # name,            coordinates, step,            scans [(seed, n_points)],  weight seed, filter
    ("egonn_cart01_b1", "cartesian", "0.1",          [(3, 12000)],              11, False),
    ("egonn_cart01_b2", "cartesian", "0.1",          [(5, 8000), (6, 6000)],    12, False),
    ("egonn_cart03_b1", "cartesian", "0.3",          [(1, 40000)],              13, True),
    ("egonn_polar_b1",  "polar",     "1., 0.3, 0.2", [(1, 40000)],              14, True),
    # BASELINE configs[1] cloud size (50 000 points, 0.1 m): the benchmark shape through the reference's own graph code
    ("egonn_cart01_50k_b2", "cartesian", "0.1",      [(100, 50000), (101, 50000)], 15, False),
]


def main():
    bootstrap_reference()
    import numpy as np
    import torch
    import MinkowskiEngine as ME
    from models.model_factory import model_factory
    from egonn_amd.synth import lidar_scan, seeded_state_dict

    torch.manual_seed(0)
    shapes_written = False
    only = sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] == "only" else None
    for name, coordinates, step, scans, wseed, filt in CASES:
        if only and name != only:
            continue
        mp = model_params(coordinates, step)
        model = model_factory(mp)
        model.eval()
        sd = model.state_dict()
        shapes = {k: [int(s) for s in v.shape] for k, v in sd.items()}
        if not shapes_written and not only:
            with open(os.path.join(HERE, "egonn_state_dict_shapes.json"), "w") as f:
                json.dump(shapes, f, indent=0, sort_keys=True)
            shapes_written = True
        new = seeded_state_dict(wseed, {k: tuple(v) for k, v in shapes.items()})
        model.load_state_dict({k: torch.from_numpy(v) for k, v in new.items()})

        out = {"weight_seed": np.int64(wseed), "coordinates": np.array(coordinates),
               "quantization_step": np.array([float(s) for s in step.split(",")], dtype=np.float64),
               "n_scans": np.int64(len(scans))}
        coords_list = []
        for b, (seed, n) in enumerate(scans):
            pc = lidar_scan(seed, n_points=n)
            if filt:
                pc = kitti_like_filter(pc)
            out[f"points_{b}"] = pc
            coords, idx = mp.quantizer(torch.from_numpy(pc))          # REFERENCE quantiser
            out[f"quant_coords_{b}"] = coords.numpy().astype(np.int32)
            out[f"quant_index_{b}"] = idx.numpy().astype(np.int64)
            coords_list.append(coords)
        bc = ME.utils.batched_coordinates(coords_list)
        feats = torch.ones((bc.shape[0], 1), dtype=torch.float32)
        with torch.no_grad():
            x = ME.SparseTensor(feats, coordinates=bc)
            levels = model.trunk(x)                                     # REFERENCE trunk
            y = model({"coords": bc, "features": feats})                # REFERENCE forward
        out["coords"] = bc.numpy().astype(np.int32)
        for lvl, t in levels.items():
            c = t.C.numpy().astype(np.int32)
            order = np.lexsort((c[:, 3], c[:, 2], c[:, 1], c[:, 0]))
            out[f"level{lvl}_coords"] = c[order]
            if lvl in (3, 7):
                out[f"level{lvl}_feats"] = t.F.numpy()[order]
        out["global"] = y["global"].numpy()
        # keypoint coordinates: rows of the local map, split exactly as the reference splits them
        xl = model.local_head(levels)
        kc = xl.C.numpy().astype(np.int32)
        for b, rows in enumerate(xl._batchwise_row_indices):
            out[f"kp_coords_{b}"] = kc[rows.numpy()]
            out[f"keypoints_{b}"] = y["keypoints"][b].numpy()
            out[f"descriptors_{b}"] = y["descriptors"][b].numpy()
            out[f"sigma_{b}"] = y["sigma"][b].numpy()
            # reference eval/evaluate.py:352-361 selection (torch.topk, ascending sigma)
            s = y["sigma"][b].squeeze(1)
            n_k = min(len(s), 128)
            _, ndx = torch.topk(s, dim=0, k=n_k, largest=False)
            out[f"topk_sigma_{b}"] = s[ndx].numpy()
            out[f"topk_coords_{b}"] = kc[rows.numpy()][ndx.numpy()]
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, "voxels", bc.shape[0], "keypoints", [len(k) for k in y["keypoints"]],
              f"{os.path.getsize(path) / 1e6:.2f} MB")

    # ---- MinkLoc3D / MinkLoc (MinkFPN backbone + GeM): reference models/minkfpn.py, models/minkloc.py,
    #      third_party/minkloc3d/minkloc.py executed on the stand-in
    if only:
        return
    for name, mname, block, step, scans, wseed in MINKLOC_CASES:
        mp = minkloc_params(mname, step, block)
        model = model_factory(mp)
        model.eval()
        sd = model.state_dict()
        shapes = {k: [int(s) for s in v.shape] for k, v in sd.items()}
        with open(os.path.join(HERE, f"{name}_state_dict_shapes.json"), "w") as f:
            json.dump(shapes, f, indent=0)                      # insertion order = reference state_dict order
        new = seeded_state_dict(wseed, {k: tuple(v) for k, v in shapes.items()})
        model.load_state_dict({k: torch.from_numpy(v) for k, v in new.items()})
        out = {"weight_seed": np.int64(wseed), "coordinates": np.array("cartesian"), "block": np.array(block),
               "model": np.array(mname), "quantization_step": np.array([float(step)]), "n_scans": np.int64(len(scans))}
        coords_list = []
        for b, (seed, n) in enumerate(scans):
            pc = kitti_like_filter(lidar_scan(seed, n_points=n))
            coords, _ = mp.quantizer(torch.from_numpy(pc))
            coords_list.append(coords)
        bc = ME.utils.batched_coordinates(coords_list)
        feats = torch.ones((bc.shape[0], 1), dtype=torch.float32)
        with torch.no_grad():
            y = model({"coords": bc, "features": feats})
            xb = model.backbone(ME.SparseTensor(feats, coordinates=bc))
        out["coords"] = bc.numpy().astype(np.int32)
        out["global"] = y["global"].numpy()
        c = xb.C.numpy().astype(np.int32)
        order = np.lexsort((c[:, 3], c[:, 2], c[:, 1], c[:, 0]))
        out["backbone_coords"] = c[order]
        out["backbone_feats"] = xb.F.numpy()[order].astype(np.float16)      # (N2, 256): stored as fp16 to stay small
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, "voxels", bc.shape[0], "backbone rows", len(c), f"{os.path.getsize(path) / 1e6:.2f} MB")


def grad_digest(name, g):
    """small, order-independent-enough summary of one gradient tensor (the full set is 4.7 M values)."""
    import zlib
    import numpy as np
    g = np.asarray(g, dtype=np.float64).reshape(-1)
    r = np.random.default_rng(zlib.crc32(name.encode())).standard_normal(g.size)
    return np.concatenate([[np.linalg.norm(g), float(g @ r)], g[:64] if g.size > 4096 else g])


TRAIN_CASES = [
    # name,               coordinates, step,  scans,                                   weight seed, projection seed
    ("egonn_train_cart03", "cartesian", "0.3", [(31, 9000), (32, 7000), (33, 8000)], 41, 42),
    ("egonn_train_polar", "polar", "1., 0.3, 0.2", [(34, 9000), (35, 8000)], 43, 44),
]


def row_weights(coords, width, salt):
    """deterministic per-row projection weights as a function of the row's (b,x,y,z) coordinate, so that both sides
    can evaluate the same linear functional of the local outputs whatever their row order is."""
    import numpy as np
    c = np.asarray(coords, dtype=np.float64)
    phase = 2.1 * c[:, 0] + 0.37 * c[:, 1] + 0.73 * c[:, 2] + 1.13 * c[:, 3] + salt
    return np.cos(phase[:, None] + 0.05 * np.arange(width)[None, :]).astype(np.float32)


def main_train():
    """train-mode step of the REFERENCE graph (models/minkgl.py in .train(): batch-statistics BatchNorm) on the
    stand-in ME ops with torch autograd: global descriptors, a linear functional of them as the loss, gradients of
    every parameter (digested) and the BatchNorm running statistics after the step."""
    bootstrap_reference()
    import numpy as np
    import torch
    import MinkowskiEngine as ME
    from models.model_factory import model_factory
    from egonn_amd.synth import lidar_scan, seeded_state_dict

    for name, coordinates, step, scans, wseed, pseed in TRAIN_CASES:
        mp = model_params(coordinates, step)
        model = model_factory(mp)
        shapes = {k: [int(s) for s in v.shape] for k, v in model.state_dict().items()}
        new = seeded_state_dict(wseed, {k: tuple(v) for k, v in shapes.items()})
        model.load_state_dict({k: torch.from_numpy(v) for k, v in new.items()})
        model.train()
        coords_list = []
        out = {"weight_seed": np.int64(wseed), "proj_seed": np.int64(pseed), "coordinates": np.array(coordinates),
               "quantization_step": np.array([float(s_) for s_ in step.split(",")]), "n_scans": np.int64(len(scans))}
        for b, (seed, n) in enumerate(scans):
            pc = kitti_like_filter(lidar_scan(seed, n_points=n))
            coords, _ = mp.quantizer(torch.from_numpy(pc))
            coords_list.append(coords)
        bc = ME.utils.batched_coordinates(coords_list)
        feats = torch.ones((bc.shape[0], 1), dtype=torch.float32)
        captured = {}
        hook = model.local_head.register_forward_hook(lambda m, i, o: captured.setdefault("xl", o))
        y = model({"coords": bc, "features": feats})                    # REFERENCE forward, train mode
        hook.remove()
        g = y["global"]
        R = torch.from_numpy(np.random.default_rng(pseed).standard_normal(tuple(g.shape)).astype(np.float32))
        loss = (g * R).sum()
        # local outputs: a linear functional with coordinate-keyed weights (row order is implementation defined)
        xl = captured["xl"]
        kc = xl.C.numpy().astype(np.int32)
        for b, rows in enumerate(xl._batchwise_row_indices):
            cb = kc[rows.numpy()]
            out[f"kp_coords_{b}"] = cb
            out[f"descriptors_{b}"] = y["descriptors"][b].detach().numpy()
            out[f"keypoints_{b}"] = y["keypoints"][b].detach().numpy()
            out[f"sigma_{b}"] = y["sigma"][b].detach().numpy()
            loss = loss + (y["descriptors"][b] * torch.from_numpy(row_weights(cb, 128, 0.1))).sum() \
                        + (y["keypoints"][b] * torch.from_numpy(row_weights(cb, 3, 0.2))).sum() \
                        + (y["sigma"][b] * torch.from_numpy(row_weights(cb, 1, 0.3))).sum()
        loss.backward()
        out["coords"] = bc.numpy().astype(np.int32)
        out["global"] = g.detach().numpy()
        out["loss"] = np.float64(loss.item())
        for k, p in model.named_parameters():
            if p.grad is not None:
                out["grad/" + k] = grad_digest(k, p.grad.numpy())
        for k, v in model.state_dict().items():
            if k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"):
                out["buf/" + k] = v.numpy()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        n_grad = sum(1 for k in out if k.startswith("grad/"))
        print(name, "voxels", bc.shape[0], "params with grad", n_grad, f"{os.path.getsize(path) / 1e6:.2f} MB")


MINKLOC_TRAIN_CASES = [
    ("minkloc3d_train_cart03", "MinkLoc3D", "BasicBlock", "0.3", [(36, 9000), (37, 8000)], 45, 46),
    ("minkloc_eca_train_cart03", "MinkLoc", "ECABasicBlock", "0.3", [(38, 9000), (39, 7000)], 47, 48),
]


def main_train_minkloc():
    """train-mode step of the reference MinkLoc3D / MinkLoc graphs (models/minkfpn.py + GeM) on the stand-in ME ops."""
    bootstrap_reference()
    import numpy as np
    import torch
    import MinkowskiEngine as ME
    from models.model_factory import model_factory
    from egonn_amd.synth import lidar_scan, seeded_state_dict

    for name, mname, block, step, scans, wseed, pseed in MINKLOC_TRAIN_CASES:
        mp = minkloc_params(mname, step, block)
        model = model_factory(mp)
        shapes = {k: [int(s) for s in v.shape] for k, v in model.state_dict().items()}
        new = seeded_state_dict(wseed, {k: tuple(v) for k, v in shapes.items()})
        model.load_state_dict({k: torch.from_numpy(v) for k, v in new.items()})
        model.train()
        coords_list = []
        for b, (seed, n) in enumerate(scans):
            pc = kitti_like_filter(lidar_scan(seed, n_points=n))
            coords, _ = mp.quantizer(torch.from_numpy(pc))
            coords_list.append(coords)
        bc = ME.utils.batched_coordinates(coords_list)
        feats = torch.ones((bc.shape[0], 1), dtype=torch.float32)
        g = model({"coords": bc, "features": feats})["global"]
        R = torch.from_numpy(np.random.default_rng(pseed).standard_normal(tuple(g.shape)).astype(np.float32))
        loss = (g * R).sum()
        loss.backward()
        out = {"weight_seed": np.int64(wseed), "proj_seed": np.int64(pseed), "model": np.array(mname),
               "block": np.array(block), "quantization_step": np.array([float(step)]), "n_scans": np.int64(len(scans)),
               "coords": bc.numpy().astype(np.int32), "global": g.detach().numpy(), "loss": np.float64(loss.item())}
        for k, p in model.named_parameters():
            if p.grad is not None:
                out["grad/" + k] = grad_digest(k, p.grad.numpy())
        for k, v in model.state_dict().items():
            if k.endswith("running_mean") or k.endswith("running_var"):
                out["buf/" + k] = v.numpy()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, "voxels", bc.shape[0], "params with grad", sum(1 for k in out if k.startswith("grad/")),
              f"{os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "train":
        main_train()
        main_train_minkloc()
        sys.exit(0)
    main()
