"""CPU-only tests: the oracle against the golden vectors captured from the reference, host-side logic (module tree,
state_dict keys, position tables, init), and that the C-ABI library loads and exports every declared symbol."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

from helpers import build_transformer, grad_floor, jload, load, rel
from oracle import fill
from oracle import vptr_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------------------- oracle pinning
@pytest.mark.parametrize("name", ["nar_tiny", "nar_tiny_norpe", "far_tiny", "nar_tiny_pad", "nar_tiny_T", "nar_tiny_tslma", "nar_tiny_tslma_pad"])
def test_oracle_matches_reference_golden(name):
    z = load(name)
    cfg, far = jload(z, "cfg"), bool(int(z["far"]))
    tmpl = jload(z, "template")
    P = fill.fill_state([tuple(t) for t in tmpl], int(z["seed"]))
    for k, shape, dt in tmpl:  # entries the fill does not produce
        if k not in P:
            if "buf:" + k in z.files:
                P[k] = torch.from_numpy(z["buf:" + k])
            elif k.endswith("relative_position_index"):
                P[k] = O.rpe_index(cfg["window_size"])
            elif k.endswith("num_batches_tracked"):
                P[k] = torch.zeros((), dtype=torch.long)
    leaves = {}
    for k, shape, dt in tmpl:
        if "grad:" + k in z.files:
            P[k] = P[k].clone().requires_grad_(True)
            leaves[k] = P[k]
    x = torch.from_numpy(z["x"]).requires_grad_(True)
    fwd = O.far_forward if far else O.nar_forward
    with torch.no_grad():
        Pe = {k: v.detach().clone() for k, v in P.items()}
        assert rel(fwd(Pe, x, cfg, training=False), z["out_eval"]) < 1e-5
    out = fwd(P, x, cfg, training=True)
    assert rel(out, z["out_train"]) < 1e-5
    (out * torch.from_numpy(z["g"])).sum().backward()
    assert rel(x.grad, z["dx"]) < 1e-5
    floor = grad_floor(np.linalg.norm(z[k]) for k in z.files if k.startswith("grad:"))
    for k, t in leaves.items():
        assert rel(t.grad, z["grad:" + k], floor) < 2e-4, k
    for k in z.files:
        if k.startswith("bn_after:"):
            assert rel(P[k[9:]], z[k]) < 1e-5


@pytest.mark.parametrize("name", ["ae_tiny_reflect", "ae_tiny_zero"])
def test_oracle_autoencoder_golden(name):
    z = load(name)
    meta = jload(z, "meta")
    Pe = fill.fill_state([tuple(t) for t in jload(z, "enc_template")], meta["seed"])
    Pd = fill.fill_state([tuple(t) for t in jload(z, "dec_template")], meta["seed"] + 10)
    x = torch.from_numpy(z["x"])
    with torch.no_grad():
        f = O.enc_forward(Pe, x, padding_type=meta["padding_type"])
    assert rel(f, z["feat"]) < 1e-5
    fin = torch.from_numpy(z["feat"]).requires_grad_(True)
    y = O.dec_forward(Pd, fin, out_layer=meta["out_layer"])
    assert rel(y, z["y"]) < 1e-5
    (y * torch.from_numpy(z["g"])).sum().backward()
    assert rel(fin.grad, z["dfeat"]) < 1e-5


def test_oracle_losses_golden():
    import torch.nn.functional as F
    z = load("losses_tiny")
    gt, gf = torch.from_numpy(z["gt"]), torch.from_numpy(z["gf"])
    pr, pf = torch.from_numpy(z["pr"]).requires_grad_(True), torch.from_numpy(z["pf"]).requires_grad_(True)
    l = O.gdl_loss(gt, pr) + O.mse_loss(pr, gt) + 0.1 * O.bipatch_nce(F.normalize(gf, p=2.0, dim=2), F.normalize(pf, p=2.0, dim=2))
    l.backward()
    assert abs(l.item() - float(z["loss"])) < 1e-6 * abs(float(z["loss"]))
    assert rel(pr.grad, z["dpr"]) < 1e-6 and rel(pf.grad, z["dpf"]) < 1e-6
    assert abs(O.mse_loss(pr, gt).item() - float(z["mse"])) < 1e-6 and abs(O.gdl_loss(gt, pr).item() - float(z["gdl"])) < 1e-6


def test_oracle_position_tables_golden():
    z = load("pos_tables")
    for tag in ("6_96_4", "20_528_4", "50_528_8"):
        T, E, ws = (int(s) for s in tag.split("_"))
        assert rel(O.pos1d(T, E), z["p1:" + tag]) < 1e-6
        assert rel(O.pos2d(E, ws, ws), z["p2:" + tag]) < 1e-6
        if "p3:" + tag in z.files:
            assert rel(O.pos3d(E, T, ws, ws), z["p3:" + tag]) < 1e-6
        else:
            n, s = fill.digest(O.pos3d(E, T, ws, ws))
            assert abs(n - float(z["p3n:" + tag])) < 1e-5 * n and rel(s, z["p3s:" + tag]) < 1e-6
    for ws in (4, 8):
        assert np.array_equal(O.rpe_index(ws).numpy(), z[f"rpe_index:{ws}"])


def test_oracle_train_step_golden():
    z = load("step_tiny")
    cfg, meta = jload(z, "cfg"), jload(z, "meta")
    Pe = fill.fill_state([tuple(t) for t in jload(z, "enc_template")], meta["seed"])
    Pd = fill.fill_state([tuple(t) for t in jload(z, "dec_template")], meta["seed"] + 10)
    tm = jload(z, "T_template")
    PT = fill.fill_state([tuple(t) for t in tm], meta["seed"] + 20)
    PT["temporal_pos"] = O.pos1d(cfg["Tp"] + cfg["Tf"], cfg["C"])
    PT["lw_pos"] = O.pos2d(cfg["C"], cfg["window_size"], cfg["window_size"])
    for k, s, d in tm:
        if k.endswith("relative_position_index"):
            PT[k] = O.rpe_index(cfg["window_size"])
        elif k.endswith("num_batches_tracked"):
            PT[k] = torch.zeros((), dtype=torch.long)
    for P, t in ((Pe, jload(z, "enc_template")), (Pd, jload(z, "dec_template"))):
        for k, s, d in t:
            if k.endswith("num_batches_tracked"):
                P[k] = torch.zeros((), dtype=torch.long)
    st = O.NARStep(Pe, Pd, PT, cfg)
    for s, ref in enumerate(jload(z, "records")):
        past = (fill.rand_input((meta["N"], cfg["Tp"], 1, meta["HW"], meta["HW"]), meta["seed"] + 100 + s) - 0.6013795) / 2.7570653
        fut = (fill.rand_input((meta["N"], cfg["Tf"], 1, meta["HW"], meta["HW"]), meta["seed"] + 200 + s) - 0.6013795) / 2.7570653
        r = st.step(past, fut)
        for k in ref:
            assert abs(r[k] - ref[k]) <= 2e-4 * abs(ref[k]) + 1e-7, (k, r[k], ref[k])
    worst = max(rel(st.P_T[k[5:]], z[k]) for k in z.files if k.startswith("post:"))
    assert worst < 1e-4


# ------------------------------------------------------------------------------------------------------ host-side logic
def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "vptr_hip.h")).read()
    declared = set(re.findall(r"\b(vptr_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"vptr_gemm_desc"}
    from vptr_amd import _lib
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.lib.vptr_abi_version() == 10
    assert ctypes.sizeof(_lib.GemmDesc) % 8 == 0


def test_graft_entry_build_runs():
    """the driver's build check (`__graft_entry__.build()`): compiles what changed, imports the package, checks the ABI version"""
    import importlib
    ge = importlib.import_module("__graft_entry__")
    ge.build()


def test_product_path_has_no_cpu_fallback():
    import vptr_amd.ops as ops
    x = torch.randn(8, 16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear(x, torch.randn(4, 16), None)
    for root, _, files in os.walk(os.path.join(ROOT, "vptr_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


@pytest.mark.parametrize("name", ["nar_tiny", "nar_tiny_norpe", "far_tiny", "nar_tiny_tslma", "nar_k64_digest", "far_bair_digest"])
def test_state_dict_keys_match_reference(name):
    import vptr_amd.model as pkg
    z = load(name)
    cfg, far = jload(z, "cfg"), bool(int(z["far"]))
    m = build_transformer(pkg, cfg, far)
    assert [(k, list(v.shape), str(v.dtype).replace("torch.", "")) for k, v in m.state_dict().items()] == \
        [tuple(t) if False else (t[0], t[1], t[2]) for t in jload(z, "template")]
    for k in ("temporal_pos", "lw_pos", "Tlw_pos"):
        if "buf:" + k in z.files:
            assert torch.equal(m.state_dict()[k], torch.from_numpy(z["buf:" + k])), k


def test_autoencoder_keys_and_api():
    import vptr_amd.model as pkg
    import model as shim
    assert shim.VPTREnc is pkg.VPTREnc and shim.VPTRFormerNAR is pkg.VPTRFormerNAR and shim.init_weights is pkg.init_weights
    for name in ("ae_tiny_reflect", "ae_tiny_zero"):
        z = load(name)
        meta = jload(z, "meta")
        enc = pkg.VPTREnc(meta["img_ch"], meta["feat"], 3, meta["padding_type"])
        dec = pkg.VPTRDec(meta["img_ch"], meta["feat"], 3, meta["out_layer"], meta["padding_type"])
        assert [(k, list(v.shape)) for k, v in enc.state_dict().items()] == [(k, s) for k, s, _ in jload(z, "enc_template")]
        assert [(k, list(v.shape)) for k, v in dec.state_dict().items()] == [(k, s) for k, s, _ in jload(z, "dec_template")]
    with pytest.raises(ValueError):
        pkg.VPTRDec(1, 48, 3, "Softmax")
    with pytest.raises(NotImplementedError):
        pkg.VPTREnc(1, 48, 3, "circular")
    enc = pkg.VPTREnc(1, 48, 3)
    pkg.init_weights(enc)
    w = enc.encoder.model[1].weight
    assert abs(float(w.std()) - 0.02) < 0.005
    assert abs(float(enc.encoder.model[2].weight.mean()) - 1.0) < 0.02


def test_reset_parameters_quirks():
    """xavier over every dim>1 parameter: RPE table, frame_queries and the 3-D LayerNorm affines are re-initialised
    (VPTR_modules.py:149-152, SURVEY.md section 8b 'Init semantics')."""
    import vptr_amd.model as pkg
    torch.manual_seed(0)
    m = pkg.VPTRFormerNAR(2, 2, 8, 8, 48, 8, 1, 1, 0.1, 4, 4, False, True)
    assert m.num_future_frames == 2 and callable(m.NCE_projector)
    ln = m.transformer.decoder.layers[0].SpatialFFN.norm1
    assert ln.weight.dim() == 3 and float(ln.weight.abs().max()) < 0.2       # not ones any more
    assert float(m.frame_queries.std()) < 0.1                                   # not randn any more
    bn = m.transformer.encoder.layers[0].SpatialFFN.norm1
    assert isinstance(bn, torch.nn.BatchNorm2d) and float(bn.weight.min()) == 1.0
    assert m.transformer.encoder.layers[0]._site != m.transformer.decoder.layers[0]._site


def test_losses_cpu_match_oracle():
    import torch.nn.functional as F
    import vptr_amd.model as pkg
    z = load("losses_tiny")
    gt, gf = torch.from_numpy(z["gt"]), torch.from_numpy(z["gf"])
    pr, pf = torch.from_numpy(z["pr"]), torch.from_numpy(z["pf"])
    N, T, C, h, w = gf.shape
    l = pkg.GDL(alpha=1)(gt, pr) + pkg.MSELoss()(pr, gt) + 0.1 * pkg.BiPatchNCE(N, T, h, w, 1.0)(
        F.normalize(gf, p=2.0, dim=2), F.normalize(pf, p=2.0, dim=2))
    assert abs(l.item() - float(z["loss"])) < 1e-6 * abs(float(z["loss"]))
    w = pkg.temporal_weight_func(10)
    assert abs(float(w[-1]) - 10.0) < 1e-4 and float(w[0]) == 1.0


# ------------------------------------------------------------------------------------------- checkpoints (section 8f rank 1)
def _tiny_modules():
    import vptr_amd.model as M
    T = M.VPTRFormerNAR(2, 2, 8, 8, 48, 8, 1, 1, 0.0, 4, 4, False, True)
    enc = M.VPTREnc(1, 48, 3, "reflect")
    return enc, T


def test_checkpoint_roundtrip_and_ddp_prefix(tmp_path):
    from vptr_amd import checkpoint as C
    enc, T = _tiny_modules()
    fill.apply_fill(enc, 1)
    fill.apply_fill(T, 2)
    opt = torch.optim.AdamW(T.parameters(), lr=1e-4)
    for p in T.parameters():
        p.grad = torch.full_like(p, 1e-3)
    opt.step()
    loss_dict = C.init_loss_dict(["T_MSE", "T_total"])
    loss_dict["T_MSE"].train.append(0.5)
    loss_dict["epochs"] = 1
    f = C.save_ckpt({"VPTR_Enc": enc, "VPTR_Transformer": T}, {"optimizer_T": opt}, 7, loss_dict, tmp_path)
    assert f.name == "epoch_7.tar"
    raw = open(f, "rb").read() if not __import__("zipfile").is_zipfile(f) else b"".join(
        __import__("zipfile").ZipFile(f).read(n) for n in __import__("zipfile").ZipFile(f).namelist() if n.endswith("data.pkl"))
    assert b"utils.train_summary" in raw and b"Loss_tuple" in raw and b"vptr_amd" not in raw  # loadable by the reference
    msd, osd, epoch, ld, code = C.load_ckpt(f)
    assert epoch == 7 and ld["T_MSE"].train == [0.5] and isinstance(ld["T_total"], C.LossTuple) and code == {}
    enc2, T2 = _tiny_modules()
    opt2 = torch.optim.AdamW(T2.parameters(), lr=3e-4)
    ld2, start = C.resume_training({"VPTR_Enc": enc2, "VPTR_Transformer": T2}, {"optimizer_T": opt2}, f, ["T_MSE", "T_total", "T_new"])
    assert start == 7 and ld2["T_new"].train == [0]
    for (k, a), (_, b) in zip(T.state_dict().items(), T2.state_dict().items()):
        assert torch.equal(a, b), k
    assert opt2.state_dict()["param_groups"][0]["lr"] == 1e-4
    # a checkpoint written under DistributedDataParallel: every key prefixed with `module.` (train_summary.py:16-21)
    ddp_sd = {"module." + k: v for k, v in T.state_dict().items()}
    torch.save({"epoch": 3, "loss_dict": {"epochs": 0}, "Module_state_dict": {"VPTR_Transformer": ddp_sd},
                "optimizer_state_dict": {}, "code": {}}, tmp_path / "ddp.tar")
    _, T3 = _tiny_modules()
    start, hist = C.resume_training({"VPTR_Transformer": T3}, {}, tmp_path / "ddp.tar", None, map_location="cpu")
    assert start == 3 and torch.equal(T3.state_dict()["frame_queries"], T.state_dict()["frame_queries"])


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference checkout (build container only)")
def test_checkpoint_interop_with_reference(tmp_path):
    """a checkpoint written by the REFERENCE's save_ckpt loads here, and one written here loads through its load_ckpt"""
    from oracle.ref_import import import_reference
    from vptr_amd import checkpoint as C
    ref = import_reference()
    import utils.train_summary as RS  # the reference's own module
    Tr = ref.VPTRFormerNAR(2, 2, 8, 8, 48, 8, 1, 1, 0.0, 4, 4, False, True)
    fill.apply_fill(Tr, 5)
    optr = torch.optim.AdamW(Tr.parameters(), lr=1e-4)
    ld = RS.init_loss_dict(["T_MSE"])
    ld["T_MSE"].val.append(0.25)
    RS.save_ckpt({"VPTR_Transformer": Tr}, {"optimizer_T": optr}, 2, ld, tmp_path / "ref")
    _, T = _tiny_modules()
    loss_dict, start = C.resume_training({"VPTR_Transformer": T}, {}, tmp_path / "ref" / "epoch_2.tar", ["T_MSE"])
    assert start == 2 and loss_dict["T_MSE"].val == [0.25]
    for (k, a), (k2, b) in zip(Tr.state_dict().items(), T.state_dict().items()):
        assert k == k2 and torch.equal(a, b), k
    f = C.save_ckpt({"VPTR_Transformer": T}, {}, 4, loss_dict, tmp_path / "ours")
    # what the reference's load_ckpt does (train_summary.py:150-160); torch >= 2.6 needs weights_only=False for its pickled class
    ck = torch.load(f, map_location=None, weights_only=False)
    assert ck["epoch"] == 4 and isinstance(ck["loss_dict"]["T_MSE"], RS.Loss_tuple) and ck["loss_dict"]["T_MSE"].val == [0.25]
    Tr.load_state_dict(ck["Module_state_dict"]["VPTR_Transformer"])


def test_flat_adamw_state_dict_is_torch_adamw_compatible():
    from vptr_amd.train import FlatAdamW
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(8, 5), torch.nn.Linear(5, 3))
    ref_opt = torch.optim.AdamW(lin.parameters(), lr=2e-4, weight_decay=0.02)
    for _ in range(3):
        for p in lin.parameters():
            p.grad = torch.randn_like(p)
        ref_opt.step()
    sd = ref_opt.state_dict()
    lin2 = torch.nn.Sequential(torch.nn.Linear(8, 5), torch.nn.Linear(5, 3))
    fo = FlatAdamW(lin2.parameters(), lr=1e-4)
    fo.load_state_dict(sd)
    assert fo.lr == 2e-4 and fo.weight_decay == 0.02 and float(fo.step_dev) == 3.0
    off = 0
    for i, p in enumerate(lin.parameters()):
        n = p.numel()
        assert torch.equal(fo.m[off:off + n].view(p.shape), sd["state"][i]["exp_avg"])
        assert torch.equal(fo.v[off:off + n].view(p.shape), sd["state"][i]["exp_avg_sq"])
        off += n
    back = torch.optim.AdamW(lin2.parameters(), lr=1.0)
    back.load_state_dict(fo.state_dict())
    b = back.state_dict()
    assert b["param_groups"][0]["lr"] == 2e-4
    for i in sd["state"]:
        assert torch.equal(b["state"][i]["exp_avg_sq"], sd["state"][i]["exp_avg_sq"]) and float(b["state"][i]["step"]) == 3.0


def test_flat_adamw_state_dict_channel_last_params():
    """Parameters that FlatAdamW STORES channel-last (LayerNorm((C,H,W)) affines, depthwise 3x3 weights) must cross the
    torch.optim.AdamW state format in their logical layout, both ways (round-1 bug: slab order leaked into the state)."""
    from vptr_amd.train import FlatAdamW
    torch.manual_seed(1)

    def make():
        return torch.nn.ModuleList([torch.nn.Conv2d(4, 4, 3, groups=4), torch.nn.LayerNorm((3, 2, 2)), torch.nn.Linear(6, 5)])
    ref = make()
    ref_opt = torch.optim.AdamW(ref.parameters(), lr=3e-4, weight_decay=0.01)
    for _ in range(2):
        for p in ref.parameters():
            p.grad = torch.randn_like(p)
        ref_opt.step()
    sd = ref_opt.state_dict()
    ours = make()
    ours.load_state_dict(ref.state_dict())
    cl = [id(ours[0].weight), id(ours[1].weight), id(ours[1].bias)]
    fo = FlatAdamW(ours.parameters(), lr=1e-4, channel_last=cl)
    for a_, b_ in zip(ours.parameters(), ref.parameters()):
        assert torch.equal(a_.detach(), b_.detach())          # re-pointing into the slab keeps the logical values
    assert not ours[0].weight.is_contiguous()                  # ... while the storage order really is channel-last
    fo.load_state_dict(sd)
    for i, p in enumerate(ours.parameters()):
        off, n, layout = fo._layout[i]
        m_logical = fo.m[off:off + n].view(layout[0]).permute(layout[1]) if layout else fo.m[off:off + n].view(p.shape)
        assert torch.equal(m_logical, sd["state"][i]["exp_avg"]), "moment of parameter %d scrambled on load" % i
        # the slab itself holds the moment in the SAME storage order as the parameter
        if layout:
            assert torch.equal(fo.m[off:off + n], sd["state"][i]["exp_avg"].permute(*[*range(1, p.dim()), 0]).reshape(-1))
    out = fo.state_dict()
    for i in sd["state"]:
        assert out["state"][i]["exp_avg"].is_contiguous()
        assert torch.equal(out["state"][i]["exp_avg"], sd["state"][i]["exp_avg"])
        assert torch.equal(out["state"][i]["exp_avg_sq"], sd["state"][i]["exp_avg_sq"])
    back = torch.optim.AdamW(make().parameters(), lr=1.0)
    back.load_state_dict(out)     # shapes are the logical ones: torch accepts the state as its own
    with pytest.raises(ValueError):
        bad = {"param_groups": sd["param_groups"], "state": {k: dict(v) for k, v in sd["state"].items()}}
        bad["state"][0]["exp_avg"] = bad["state"][0]["exp_avg"].reshape(-1)
        fo.load_state_dict(bad)


def test_oracle_far_train_step_golden():
    z = load("step_far_tiny")
    cfg, meta = jload(z, "cfg"), jload(z, "meta")
    import vptr_amd.model as M
    enc = M.VPTREnc(1, meta["feat"], 3, "reflect")
    dec = M.VPTRDec(1, meta["feat"], 3, meta["out_layer"], "reflect")
    T = build_transformer(M, cfg, True)
    fill.apply_fill(enc, meta["seed"])
    fill.apply_fill(dec, meta["seed"] + 10)
    fill.apply_fill(T, meta["seed"] + 20)
    st = O.FARStep(dict(enc.state_dict()), dict(dec.state_dict()), dict(T.state_dict()), cfg, out_layer=meta["out_layer"])
    for s, ref in enumerate(jload(z, "records")):
        past = fill.rand_input((meta["N"], cfg["Tp"], 1, meta["HW"], meta["HW"]), meta["seed"] + 100 + s)
        fut = fill.rand_input((meta["N"], cfg["Tf"], 1, meta["HW"], meta["HW"]), meta["seed"] + 200 + s)
        r = st.step(past, fut)
        for k in ref:
            assert abs(r[k] - ref[k]) <= 2e-4 * abs(ref[k]) + 1e-7, (k, r[k], ref[k])
    worst = max(rel(st.P_T[k[5:]], z[k]) for k in z.files if k.startswith("post:"))
    assert worst < 1e-4


def test_oracle_ae_gan_step_golden():
    z = load("step_ae_tiny")
    meta = jload(z, "meta")
    import vptr_amd.model as M
    enc = M.VPTREnc(meta["cimg"], meta["feat"], 3, "reflect")
    dec = M.VPTRDec(meta["cimg"], meta["feat"], 3, "Tanh", "reflect")
    disc = M.VPTRDisc(meta["cimg"], ndf=64, n_layers=3)
    fill.apply_fill(enc, meta["seed"])
    fill.apply_fill(dec, meta["seed"] + 10)
    fill.apply_fill(disc, meta["seed"] + 20)
    st = O.AEStep(dict(enc.state_dict()), dict(dec.state_dict()), dict(disc.state_dict()), lam_gan=meta["lam_gan"])
    shape = (meta["N"], meta["T"], meta["cimg"], meta["HW"], meta["HW"])
    for s, ref in enumerate(jload(z, "records")):
        past = (fill.rand_input(shape, meta["seed"] + 100 + s) - 0.6013795) / 2.7570653
        fut = (fill.rand_input(shape, meta["seed"] + 200 + s) - 0.6013795) / 2.7570653
        r = st.step(past, fut)
        for k in ref:
            assert abs(r[k] - ref[k]) <= 2e-4 * abs(ref[k]) + 1e-7, (k, r[k], ref[k])
    P = {"enc": st.P_enc, "dec": st.P_dec, "disc": st.P_disc}
    for k in z.files:
        if k.startswith("post:"):
            _, tag, name = k.split(":", 2)
            flat = P[tag][name].detach().flatten()
            assert rel(flat[::max(1, flat.numel() // 4096)], z[k]) < 1e-4, k


def test_metrics_match_reference_golden():
    from vptr_amd import metrics as M
    z = load("metrics_tiny")
    exp = jload(z, "expected")
    for tag, e in exp.items():
        x, y = torch.from_numpy(z["x:" + tag]), torch.from_numpy(z["y:" + tag])
        assert abs(M.PSNR(x, y) - e["psnr"]) < 1e-4 and abs(M.PSNR(x * 255, y * 255, 255) - e["psnr255"]) < 1e-4
        assert abs(M.MSEScore(x, y) - e["mse"]) < 1e-5 * abs(e["mse"])
        assert abs(float(M.SSIM()(x, y)) - e["ssim"]) < 2e-6
        assert np.allclose(M.SSIM(size_average=False)(x, y).numpy(), np.array(e["ssim_each"]), atol=2e-6)


def test_oracle_nar_gan_step_golden():
    z = load("step_nar_gan_tiny")
    cfg, meta = jload(z, "cfg"), jload(z, "meta")
    import vptr_amd.model as M
    enc = M.VPTREnc(1, meta["feat"], 3, "reflect")
    dec = M.VPTRDec(1, meta["feat"], 3, "Tanh", "reflect")
    disc = M.VPTRDisc(1, ndf=64, n_layers=3)
    T = build_transformer(M, cfg, False)
    fill.apply_fill(enc, meta["seed"])
    fill.apply_fill(dec, meta["seed"] + 10)
    fill.apply_fill(T, meta["seed"] + 20)
    fill.apply_fill(disc, meta["seed"] + 30)
    st = O.NARStep(dict(enc.state_dict()), dict(dec.state_dict()), dict(T.state_dict()), cfg, P_disc=dict(disc.state_dict()),
                   lam_gan=meta["lam_gan"])
    for s, ref in enumerate(jload(z, "records")):
        past = (fill.rand_input((meta["N"], cfg["Tp"], 1, meta["HW"], meta["HW"]), meta["seed"] + 100 + s) - 0.6013795) / 2.7570653
        fut = (fill.rand_input((meta["N"], cfg["Tf"], 1, meta["HW"], meta["HW"]), meta["seed"] + 200 + s) - 0.6013795) / 2.7570653
        r = st.step(past, fut)
        for k in ref:
            assert abs(r[k] - ref[k]) <= 2e-4 * abs(ref[k]) + 1e-7, (k, r[k], ref[k])


def test_rollout_loops_match_reference_notebook_on_cpu():
    """The rollout LOOPS of vptr_amd.inference (model-agnostic host logic) driven with oracle-backed callables vs the tensors the
    reference's own notebook functions produced (tests/golden/rollouts_tiny.npz, see oracle/make_golden.py::rollout_case)."""
    from oracle import vptr_oracle as O
    from vptr_amd.inference import far_rollout, nar_bair_2_to_28, nar_rollout
    import vptr_amd.model as pkg
    z = load("rollouts_tiny")
    meta = jload(z, "meta")

    class Fn:
        def __init__(self, f, **attrs):
            self.f = f
            self.__dict__.update(attrs)

        def eval(self):
            return self

        def __call__(self, x):
            return self.f(x)

    def state(module, seed):
        fill.apply_fill(module, seed)
        return {k: v.detach().clone() for k, v in module.state_dict().items()}
    Pe = state(pkg.VPTREnc(1, meta["feat"], 3, "reflect"), meta["seed"])
    Pd = state(pkg.VPTRDec(1, meta["feat"], 3, "Sigmoid", "reflect"), meta["seed"] + 1)
    enc = Fn(lambda x: O.enc_forward(Pe, x))
    dec = Fn(lambda f: O.dec_forward(Pd, f, out_layer="Sigmoid"))
    cfg = jload(z, "cfg_far")
    Pf = state(build_transformer(pkg, cfg, True), meta["seed"] + 2)
    far = Fn(lambda f: O.far_forward(Pf, f, cfg), num_future_frames=cfg["Tf"])
    past = torch.from_numpy(z["far_past"])
    for mode, key in (("RIP", "far_rip"), ("RIL", "far_ril")):
        got = far_rollout(enc, dec, far, past, z[key].shape[1], mode=mode)
        assert got.shape == tuple(z[key].shape) and rel(got, z[key]) < 2e-5, (mode, rel(got, z[key]))
    cfgn = jload(z, "cfg_nar")
    Pn = state(build_transformer(pkg, cfgn, False), meta["seed"] + 5)
    nar = Fn(lambda f: O.nar_forward(Pn, f, cfgn), num_future_frames=cfgn["Tf"])
    got = nar_rollout(enc, dec, nar, torch.from_numpy(z["nar_past"]), rounds=2, chain="feats")
    assert rel(got, z["nar_chained"]) < 2e-5
    cfgb = jload(z, "cfg_bair")
    Pb = state(build_transformer(pkg, cfgb, False), meta["seed"] + 8)
    narb = Fn(lambda f: O.nar_forward(Pb, f, cfgb), num_future_frames=cfgb["Tf"])
    got = nar_bair_2_to_28(enc, dec, narb, torch.from_numpy(z["bair_past"]))
    assert got.shape == tuple(z["bair_frames"].shape) and rel(got, z["bair_frames"]) < 2e-5
    with pytest.raises(ValueError):
        nar_rollout(enc, dec, narb, torch.from_numpy(z["bair_past"]), rounds=2, chain="feats")   # Tf != Tp cannot chain features


def test_bench_refuses_fewer_devices_than_asked():
    """`python bench.py --gpus 2` on a host with fewer than 2 visible GPUs (this container has none) must exit non-zero with a clear
    message and print no result line -- never a silent one-GPU measurement labelled n_gpus 1"""
    import subprocess
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("this host can run --gpus 2")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VPTR_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "GPU(s) visible" in (r.stderr + r.stdout) and "--gpus 2" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    # a launcher that brought up a different number of ranks than --gpus says is refused too, in either direction
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in (r.stderr + r.stdout)


@pytest.mark.parametrize("rows_mode", [128, 256, 192])
@pytest.mark.parametrize("allow_sync", [True, False])
def test_wgrad_launch_plan_partitions_every_problem(rows_mode, allow_sync):
    """host logic of the grouped weight-gradient flush (vptr_amd.ops.plan_wgrad_launches, pure): whatever tile geometry is chosen, the
    sub-problems of the launches cover every (row, token) of every weight's gradient exactly once, with consistent operand / destination /
    bias-gradient addresses; a launch vouched for the panel-synchronous kernel walks one token count; the column sums of a flipped
    problem land in exactly one row range whose last tile has a free 16-row fragment."""
    from vptr_amd import ops
    F, C = 2112, 528
    shapes = [(F, C), (C, F), (C, C), (3 * C, C), (C, C), (F, C)]          # (N, K) of dW = dY^T X: fc1, fc2, projections, packed in_proj, ...
    probs, meta = [], {}
    addr = 1 << 20
    for li, tokens in enumerate((10240, 10240, 5120)):                      # two token counts: encoder / decoder layers of KTH128
        for (N, K) in shapes:
            flip = N < K and N % 176 == 0 and K % 128 != 0 and 128 - K % 128 >= 32      # the rule of ops._launch_wgrad_group
            rows, cols = (K, N) if flip else (N, K)
            ap, bp, dp, rp = addr, addr + (1 << 30), addr + (2 << 30), addr + (3 << 30)
            addr += 1 << 24
            lda, ldb, ldd = rows, cols, K                                    # token-major operands, dW [N][K]
            probs.append((ap, bp, dp, rp, lda, ldb, ldd, rows, cols, tokens, 1.0, flip))
            meta[dp] = (ap, bp, dp, rp, lda, ldb, ldd, rows, cols, tokens, flip)
    launches = ops.plan_wgrad_launches(probs, 176, True, 1, allow_sync, rows_mode)
    assert launches
    cover = {dp: [] for dp in meta}
    for subs, vouch in launches:
        trs = {(s[12] if len(s) > 12 else 128) for s in subs}
        assert len(trs) == 1, "one tile geometry per launch"
        tr = trs.pop()
        if vouch:
            assert allow_sync and len({s[9] for s in subs}) == 1, "a vouched launch walks one token count"
        for s in subs:
            (ap, bp, dp, rp, lda, ldb, ldd, rows, cols, tokens, alpha, flip) = s[:12]
            owner = max(k for k in meta if k <= dp)
            (ap0, bp0, dp0, rp0, lda0, ldb0, ldd0, rows0, cols0, tokens0, flip0) = meta[owner]
            assert (lda, ldb, ldd, cols, flip) == (lda0, ldb0, ldd0, cols0, flip0)
            r0 = ((dp - dp0) // 4) if flip else ((dp - dp0) // (4 * ldd))
            assert dp - dp0 == (r0 * 4 if flip else r0 * ldd * 4)
            da = ap - ap0                                                    # = token offset * lda * 4 + row offset * 4
            t0, rr = divmod(da // 4, lda)
            assert rr == r0 and da % 4 == 0 and bp - bp0 == t0 * ldb * 4
            assert 0 <= r0 and r0 + rows <= rows0 and 0 <= t0 and t0 + tokens <= tokens0 and t0 % 32 == 0
            if rp:
                assert rp == rp0 + (0 if flip else r0 * 4)
                if flip:   # column sums: the tile that holds the last rows of this range must have a free 16-row fragment
                    assert r0 + rows == rows0 and ((rows + tr - 1) // tr) * tr - rows >= 16, (rows, tr)
            elif not flip:
                raise AssertionError("a row range of a non-flipped problem lost its bias gradient")
            cover[owner].append((r0, rows, t0, tokens, bool(rp)))
    for dp0, pieces in cover.items():
        rows0, tokens0, flip0 = meta[dp0][7], meta[dp0][9], meta[dp0][10]
        assert sum(r * t for (_, r, _, t, _) in pieces) == rows0 * tokens0, "rows x tokens not covered exactly once"
        cells = set()
        for (r0, r, t0, t, _) in pieces:
            key = (r0, r, t0, t)
            assert key not in cells
            cells.add(key)
        row_ranges = sorted({(r0, r) for (r0, r, _, _, _) in pieces})
        assert row_ranges[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(row_ranges, row_ranges[1:])) and sum(r for _, r in row_ranges) == rows0
        for (r0, r) in row_ranges:
            toks = sorted((t0, t) for (a, b, t0, t, _) in pieces if (a, b) == (r0, r))
            assert toks[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(toks, toks[1:])) and sum(t for _, t in toks) == tokens0
        if flip0:   # exactly one row range carries the column sums (once per token range)
            assert len({(r0, r) for (r0, r, _, _, has) in pieces if has}) == 1


def test_wgrad_launch_plan_cuts_small_groups_into_token_ranges():
    """one layer's weight (the launches of a torch.distributed job / torch.autograd.grad): 15 tiles become ~500 by token ranges of >= 1024
    tokens, all accumulating into the same destination"""
    from vptr_amd import ops
    C, tokens = 528, 10240
    prob = (1 << 20, 1 << 30, 2 << 30, 3 << 30, C, C, C, C, C, tokens, 1.0, False)
    (subs, vouch), = ops.plan_wgrad_launches([prob], 176, True, 1, True, 256)
    assert len(subs) == 10 and not any(len(s) > 12 and s[12] != 128 for s in subs)
    assert all(s[2] == prob[2] and s[3] == prob[3] for s in subs) and sum(s[9] for s in subs) == tokens and min(s[9] for s in subs) >= 1024
    (subs1, _), = ops.plan_wgrad_launches([prob], 176, True, 1, True, 256, token_split=False)
    assert len(subs1) == 1


def test_analytic_zero_class_is_an_allow_list():
    """every gradient the norm rule of helpers.analytic_zero classes as zero in a committed fixture carries a name from the explicit
    allow-list: a small-but-real gradient can not slip into the class"""
    import glob
    import json
    import numpy as np
    from helpers import analytic_zero
    classed = 0
    for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz"))):
        z = np.load(f, allow_pickle=False)
        if "grad_norms" not in z.files:
            continue
        gn = json.loads(str(z["grad_norms"]))
        classed += sum(analytic_zero(gn[k], gn.values(), k) for k in gn)   # asserts inside on a name outside the list
    assert classed > 0


def test_winograd_filter_transform_matches_direct_convolution():
    """host side of the frozen encoder's Winograd path (vptr_amd/ops/conv.py::_WINO_G, the G of F(4x4, 3x3)) against the kernels' B^T and A^T
    (csrc/winograd.hip wino_bt6 / wino_at4, restated here): A^T [(G g G^T) . (B^T d B)] A == the direct 3 x 3 convolution of a 6 x 6 patch"""
    from vptr_amd.ops.conv import _WINO_G
    G = torch.tensor(_WINO_G, dtype=torch.float64)
    BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                       [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
    AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)
    g = torch.randn(5, 3, 3, 3, dtype=torch.float64)          # [Cout, Cin, 3, 3]
    d = torch.randn(1, 3, 6, 6, dtype=torch.float64)
    U = torch.einsum("ia,kcab,jb->ijkc", G, g, G)             # as ops.wino_filter
    V = torch.einsum("ij,cjk,lk->ilc", BT, d[0], BT)          # B^T d B per channel
    M = torch.einsum("ijkc,ijc->ijk", U, V)
    Y = torch.einsum("ai,ijk,bj->kab", AT, M, AT)             # [Cout, 4, 4]
    ref = torch.nn.functional.conv2d(d, g)[0]
    assert float((Y - ref).abs().max()) < 1e-12
