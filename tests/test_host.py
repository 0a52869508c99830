"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, the module
tree reproduces the reference state_dict, the INI surface parses, and the product path refuses to run
without a HIP device (no CPU fallback)."""
import json
import os
import re

import numpy as np
import pytest
import torch

from tests import helpers as H

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    return True


def test_library_exports_every_declared_symbol(built):
    from egonn_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(REPO, "include", "egonn_hip.h")).read()
    declared = set(re.findall(r"\b(egonn_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_state_dict_matches_reference_keys_shapes_and_order(built):
    from egonn_amd import ModelParams, model_factory
    m = model_factory(ModelParams(model="egonn", coordinates="cartesian", quantization_step=0.1))
    sd = m.state_dict()
    ref = H.state_dict_shapes()
    assert set(sd.keys()) == set(ref.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == ref[k], k
    assert sum(p.nelement() for p in m.parameters()) == 4707298
    assert sum(p.nelement() for p in m.trunk.parameters()) == 4253821
    # load_state_dict(strict) round trip with seeded weights
    w = H.seeded_weights(3)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)


def test_model_params_ini(tmp_path, built):
    from egonn_amd import ModelParams, CartesianQuantizer, PolarQuantizer
    p = tmp_path / "egonn.txt"
    p.write_text("[MODEL]\nmodel = egonn\ncoordinates = polar\nquantization_step = 1., 0.3, 0.2\n")
    mp = ModelParams(str(p))
    assert mp.model == "egonn" and isinstance(mp.quantizer, PolarQuantizer)
    assert mp.quantization_step == [1.0, 0.3, 0.2] and mp.quantizer.theta_range == 360
    p.write_text("[MODEL]\nmodel = egonn\ncoordinates = cartesian\nquantization_step = 0.1\n")
    mp = ModelParams(str(p))
    assert isinstance(mp.quantizer, CartesianQuantizer) and mp.quantization_step == 0.1


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback(built):
    from egonn_amd import ModelParams, model_factory
    m = model_factory(ModelParams(model="egonn", coordinates="cartesian", quantization_step=0.1)).eval()
    batch = {"coords": torch.zeros((1, 4), dtype=torch.int32), "features": torch.ones((1, 1))}
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(batch)
    with pytest.raises(RuntimeError, match="no CPU fallback|no HIP device"):
        m.quantizer(torch.zeros((4, 3)))


def test_training_mode_has_no_cpu_path_either(built):
    from egonn_amd import ModelParams, model_factory
    m = model_factory(ModelParams(model="egonn", coordinates="cartesian", quantization_step=0.1)).train()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m({"coords": torch.zeros((1, 4), dtype=torch.int32), "features": torch.ones((1, 1))})


def test_local_losses_have_no_cpu_path_and_fixture_is_complete(built):
    """the local-head losses (reference models/loss_utils.py) run only on the HIP device; their golden vectors (made by
    importing the reference, tests/golden/make_golden_losses.py) carry every array the GPU tests read."""
    import os
    from egonn_amd import KeypointLoss, CorrespondenceLoss
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "local_losses.npz"))
    for name in fx["cases"]:
        for k in ("pc1", "pc2", "kp1", "kp2", "sigma1", "sigma2", "desc1", "desc2", "M", "dist", "loss_keypoint", "loss_correspondence",
                  "loss_total", "f64_loss_total", "leaf_grad_dist", "grad_kp1", "f64_grad_desc2", "f64_leaf_grad_kp2",
                  "metric_matching_keypoints", "f64_metric_repeatability"):
            assert f"{name}_{k}" in fx.files, (name, k)
        assert fx[f"{name}_dist"].shape == (len(fx[f"{name}_kp1"]), len(fx[f"{name}_kp2"]))
        assert np.isfinite(fx[f"{name}_loss_total"]) and abs(float(fx[f"{name}_loss_total"]) - float(fx[f"{name}_f64_loss_total"])) < 1e-3
    if not torch.cuda.is_available():
        a = {k: torch.from_numpy(fx[f"a_{k}"]) for k in ("pc1", "pc2", "kp1", "kp2", "sigma1", "sigma2", "desc1", "desc2", "dist")}
        with pytest.raises(RuntimeError, match="HIP device"):
            KeypointLoss()(a["pc1"], a["kp1"], a["sigma1"], a["pc2"], a["kp2"], a["sigma2"], a["dist"])
        with pytest.raises(RuntimeError, match="HIP device"):
            CorrespondenceLoss(beta=2.0)(a["desc1"], a["desc2"], a["dist"])


def test_synthetic_generator_is_deterministic():
    from egonn_amd.synth import lidar_scan
    a, b = lidar_scan(4, 5000), lidar_scan(4, 5000)
    assert a.shape == (5000, 3) and a.dtype.name == "float32" and (a == b).all()
    case = H.load_case("egonn_cart01_b1")
    assert (lidar_scan(3, 12000) == case["points_0"]).all()      # fixtures are reproducible from the seed


@pytest.mark.parametrize("name,kind", [("minkloc3d_cart03_b2", "MinkLoc3D"), ("minkloc_eca_cart03", "MinkLoc")])
def test_minkloc_state_dict_matches_reference(built, name, kind):
    """MinkLoc3D / MinkLoc(ECABasicBlock): same keys, shapes AND order as the reference modules."""
    import egonn_amd
    if kind == "MinkLoc3D":
        m = egonn_amd.model_factory(egonn_amd.ModelParams(model="MinkLoc3D", coordinates="cartesian", quantization_step=0.3))
    else:
        m = egonn_amd.model_factory(egonn_amd.ModelParams(model="MinkLoc", coordinates="cartesian", quantization_step=0.3,
                                                         block="ECABasicBlock", planes="32,64,64", layers="1,1,1"))
    sd = m.state_dict()
    ref = H.state_dict_shapes(name)
    assert list(sd.keys()) == list(ref.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == ref[k], k


def test_status_codes_map_to_exception_types(built):
    """C status 5 (EGONN_STATUS_CAPACITY: a batch did not fit egonn_ctx_reserve) raises CapacityError — what the streaming
    pipeline catches to fall back to the exact-size eager path; every other non-zero status raises EgonnError; both are
    RuntimeErrors (callers written against round <= 3 keep working) and carry the C code.  The header documents the codes."""
    from egonn_amd import _lib
    assert issubclass(_lib.CapacityError, _lib.EgonnError) and issubclass(_lib.EgonnError, RuntimeError)
    _lib.check(0)
    with pytest.raises(_lib.CapacityError) as e:
        _lib.check(5)
    assert e.value.code == 5
    with pytest.raises(_lib.EgonnError) as e:
        _lib.check(3)
    assert e.value.code == 3 and not isinstance(e.value, _lib.CapacityError)
    header = open(os.path.join(REPO, "include", "egonn_hip.h")).read()
    assert "EGONN_STATUS_CAPACITY = 5" in header and "EGONN_STATUS_RANGE = 3" in header
