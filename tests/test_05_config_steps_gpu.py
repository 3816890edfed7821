"""Two TRAIN STEPS at the literal size of every BASELINE.json configuration, against records the REFERENCE itself produced
(oracle/make_golden.py imports /root/reference, runs `single_iter` of the matching script for two iterations on RandomState-filled
weights and synthetic clips in the data pipeline's value range, asserts the oracle agrees, and stores the loss terms, the gradient
norm and a strided sample of every post-step parameter tensor):

  config 1  stage-1 auto-encoder + PatchGAN on MovingMNIST, feat 528, batch 4 x 20 frames, Sigmoid decoder on raw [0, 1) frames
            (train_AutoEncoder.py:44-86,121-139; ResNetAutoEncoder.py:91-96)                          step_ae528_mnist_digest
  config 2  MovingMNIST NAR 10 -> 10, 4 + 8 layers, Sigmoid decoder (train_NAR.py:49-107; Test_VPTR.ipynb cell 3)  step_mnist_digest
  config 3  KTH NAR 10 -> 10 (the bench workload): tests/test_02_model_gpu.py::test_k64_train_step_digest (batch 1 and 4)
  config 4  BAIR FAR 2 -> 28: VPTRFormerFAR(2, 28, 12 layers, RPE), T_in = 29, 3-channel frames, zero padding, BAIR normalisation
            (train_FAR.py:48-101; train_FAR_mp.py:289-300; utils/dataset.py:47-50)                     step_bair29_digest
  config 5  KTH 128 x 128 10 -> 40: 16 x 16 feature maps, 8 x 8 windows, 152.6 M parameters           step_kth128_digest
  round 5   configs 4 and 5 again at batch sizes that run multi-round GEMM grids: BAIR N = 6 (11 136 tokens), KTH128 N = 2 (the bench's
            per-GPU batch, 20 480 tokens; two panel-synchronous weight-gradient launches)   step_bair29_n6_digest, step_kth128_n2_digest

Bars: loss terms 1e-3 (north_star), gradient norm 2e-3, post-step parameters 2e-4 rel-L2 over the sampled elements with at most 3 %
of the sampled updates off by more than lr / 2 (the first AdamW updates are ~lr * sign(g): helpers.sampled_post_params_close)."""
import pytest
import torch

from helpers import build_transformer, jload, load, sampled_post_params_close
from oracle import fill

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module")
def pkg():
    import vptr_amd.model as M
    return M


def _release():
    from vptr_amd import ops
    ops.unregister_flat_slabs()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("fixture", ["step_mnist_digest", "step_kth128_digest", "step_kth128_n2_digest"])
def test_nar_config_step_digest(pkg, dev, fixture):
    from vptr_amd.train import NARTrainer
    z = load(fixture)
    cfg, meta = jload(z, "cfg"), jload(z, "meta")
    out_layer, norm = meta.get("out_layer", "Tanh"), meta.get("norm", "kth")
    enc = pkg.VPTREnc(1, meta["feat"], 3, "reflect")
    dec = pkg.VPTRDec(1, meta["feat"], 3, out_layer, "reflect")
    T = build_transformer(pkg, cfg, False)
    fill.apply_fill(enc, meta["seed"])
    fill.apply_fill(dec, meta["seed"] + 10)
    fill.apply_fill(T, meta["seed"] + 20)
    tr = NARTrainer(enc.to(dev), dec.to(dev), T.to(dev), batch_size=meta["N"], lr=1e-4, max_grad_norm=1.0, lam_pc=0.1)
    for s, ref in enumerate(jload(z, "records")):
        past = fill.clip_input((meta["N"], cfg["Tp"], 1, meta["HW"], meta["HW"]), meta["seed"] + 100 + s, norm).to(dev)
        fut = fill.clip_input((meta["N"], cfg["Tf"], 1, meta["HW"], meta["HW"]), meta["seed"] + 200 + s, norm).to(dev)
        out = tr.step(past, fut)
        for k in ("T_total", "T_GDL", "T_MSE", "T_bpc"):
            assert abs(float(out[k]) - ref[k]) < TOL * abs(ref[k]) + 1e-6, (s, k, float(out[k]), ref[k])
        assert abs(float(out["grad_norm"]) - ref["grad_norm"]) < 2e-3 * ref["grad_norm"], (s, float(out["grad_norm"]), ref["grad_norm"])
    sampled_post_params_close({"T": T.state_dict()}, z, lr=1e-4, rel_tol=2e-4)
    del tr
    _release()


@pytest.mark.parametrize("fixture", ["step_bair29_digest", "step_bair29_n6_digest"])
def test_far_bair29_step_digest(pkg, dev, fixture):
    from vptr_amd.train import FARTrainer
    z = load(fixture)
    cfg, meta = jload(z, "cfg"), jload(z, "meta")
    assert meta["cimg"] == 3 and meta["padding_type"] == "zero" and cfg["Tp"] + cfg["Tf"] - 1 == 29
    enc = pkg.VPTREnc(meta["cimg"], meta["feat"], 3, meta["padding_type"])
    dec = pkg.VPTRDec(meta["cimg"], meta["feat"], 3, meta["out_layer"], meta["padding_type"])
    T = build_transformer(pkg, cfg, True)
    fill.apply_fill(enc, meta["seed"])
    fill.apply_fill(dec, meta["seed"] + 10)
    fill.apply_fill(T, meta["seed"] + 20)
    tr = FARTrainer(enc.to(dev), dec.to(dev), T.to(dev), lr=1e-4, max_grad_norm=1.0)
    for s, ref in enumerate(jload(z, "records")):
        past = fill.clip_input((meta["N"], cfg["Tp"], meta["cimg"], meta["HW"], meta["HW"]), meta["seed"] + 100 + s, meta["norm"]).to(dev)
        fut = fill.clip_input((meta["N"], cfg["Tf"], meta["cimg"], meta["HW"], meta["HW"]), meta["seed"] + 200 + s, meta["norm"]).to(dev)
        out = tr.step(past, fut)
        for k in ("T_total", "T_GDL", "T_MSE"):
            assert abs(float(out[k]) - ref[k]) < TOL * abs(ref[k]) + 1e-6, (s, k, float(out[k]), ref[k])
        assert abs(float(out["grad_norm"]) - ref["grad_norm"]) < 2e-3 * ref["grad_norm"], (s, float(out["grad_norm"]), ref["grad_norm"])
    sampled_post_params_close({"T": T.state_dict()}, z, lr=1e-4, rel_tol=2e-4)
    del tr
    _release()


def test_ae528_mnist_step_digest(pkg, dev):
    from vptr_amd.train import AETrainer
    z = load("step_ae528_mnist_digest")
    meta = jload(z, "meta")
    assert meta["feat"] == 528 and meta["N"] == 4 and meta["T"] == 10 and meta["out_layer"] == "Sigmoid"
    enc = pkg.VPTREnc(meta["cimg"], meta["feat"], 3, "reflect")
    dec = pkg.VPTRDec(meta["cimg"], meta["feat"], 3, meta["out_layer"], "reflect")
    disc = pkg.VPTRDisc(meta["cimg"], ndf=64, n_layers=3)
    fill.apply_fill(enc, meta["seed"])
    fill.apply_fill(dec, meta["seed"] + 10)
    fill.apply_fill(disc, meta["seed"] + 20)
    tr = AETrainer(enc.to(dev), dec.to(dev), disc.to(dev), lr=2e-4, lam_gan=meta["lam_gan"])
    shape = (meta["N"], meta["T"], meta["cimg"], meta["HW"], meta["HW"])
    for s, ref in enumerate(jload(z, "records")):
        past = fill.clip_input(shape, meta["seed"] + 100 + s, meta["norm"]).to(dev)
        fut = fill.clip_input(shape, meta["seed"] + 200 + s, meta["norm"]).to(dev)
        out = tr.step(past, fut)
        for k, v in ref.items():
            assert abs(float(out[k]) - v) < 2 * TOL * abs(v) + 1e-6, (s, k, float(out[k]), v)
    sampled_post_params_close({"enc": enc.state_dict(), "dec": dec.state_dict(), "disc": disc.state_dict()}, z, lr=2e-4, rel_tol=4e-4)
    del tr
    _release()
