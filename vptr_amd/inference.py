"""Test-time rollouts of VPTR on the MI355X path.

The reference defines them in Test_VPTR.ipynb cell 5 and train_FAR.py:103-125; these are the loops behind the "10 -> 40" and
"2 -> 28" settings of BASELINE.json configs 4 / 5:

`nar_rollout(chain="feats")` -- NAR_test_single_iter: round r+1 takes the PREDICTED FEATURES of round r as its past (no
                                Dec -> Enc round trip); needs Tf == Tp from the second round on.
`nar_rollout(chain="frames")`-- every round re-encodes the last Tp predicted frames (the scheme of the BAIR function below for an
                                arbitrary number of rounds).
`nar_bair_2_to_28`           -- NAR_BAIR_2_to_28_test_single_iter: three re-encoded rounds from 2 frames, the third round
                                trimmed by two frames (10 + 10 + 8 = 28 with the released Tf = 10 model).
`far_rollout(mode="RIP")`    -- FAR_RIP_test_single_iter: recurrent inference over PIXELS -- the newest predicted feature is decoded
                                and re-encoded before it joins the input window; once `num_future_frames` predictions have been
                                appended the window slides (its oldest feature is dropped).
`far_rollout(mode="RIL")`    -- FAR_RIL_test_single_iter: the same over the LATENT space (predicted features are appended as-is).
`far_rollout(mode="train")`  -- FAR_show_sample's test phase (train_FAR.py:103-125): growing window, Dec -> Enc from the second
                                prediction on, ONE decoder pass at the end; also returns the re-predicted past frames.

Everything runs under no_grad through the same HIP kernels as training (eval mode: dropout / DropPath off).
"""
import torch


@torch.no_grad()
def nar_rollout(enc, dec, T, past, rounds=1, chain="feats"):
    """past (N,Tp,C,H,W) -> predicted frames (N, rounds*Tf, C, H, W)."""
    if chain not in ("feats", "frames"):
        raise ValueError("nar_rollout: chain must be 'feats' or 'frames'")
    T.eval()
    Tp = past.shape[1]
    out = []
    feats = enc(past)
    for r in range(rounds):
        pred_feats = T(feats)
        pred = dec(pred_feats)
        out.append(pred)
        if r + 1 == rounds:
            break
        if chain == "feats":
            if pred_feats.shape[1] != Tp:
                raise ValueError("nar_rollout(chain='feats'): chained rounds need num_future_frames == num_past_frames "
                                 "(got %d -> %d)" % (Tp, pred_feats.shape[1]))
            feats = pred_feats                      # past_gt_feats = pred_future_feats
        else:
            if pred.shape[1] < Tp:
                raise ValueError("nar_rollout(chain='frames'): a round predicts fewer frames than the model's past length")
            feats = enc(pred[:, -Tp:])
    return torch.cat(out, dim=1)


@torch.no_grad()
def nar_bair_2_to_28(enc, dec, T, past):
    """past (N,2,C,H,W) -> (N, 3*Tf - 2, C, H, W): rounds 2 and 3 start from the last two predicted frames, and the last two
    frames of round 3 are dropped (28 = 10 + 10 + 8 for Tf = 10)."""
    if past.shape[1] != 2:
        raise ValueError("nar_bair_2_to_28: expects two past frames")
    T.eval()
    out = []
    cur = past
    for r in range(3):
        pred = dec(T(enc(cur)))
        out.append(pred if r < 2 else pred[:, :-2])
        cur = pred[:, -2:]
    return torch.cat(out, dim=1)


@torch.no_grad()
def far_rollout(enc, dec, T, past, num_pred, mode="train"):
    """mode 'RIP' / 'RIL': past (N,Tp,C,H,W) -> predicted future frames (N,num_pred,C,H,W).
    mode 'train': -> (pred_past_frames (N,Tp-1,...), pred_future_frames (N,num_pred,...)) exactly as FAR_show_sample's test phase."""
    T.eval()
    past_feats = enc(past)
    pred_feats = T(past_feats)
    if mode == "train":
        input_feats = past_feats
        for i in range(num_pred - 1):
            if i == 0:
                input_feats = torch.cat([past_feats, pred_feats[:, -1:]], dim=1)
            else:
                input_feats = torch.cat([input_feats, enc(dec(pred_feats[:, -1:]))], dim=1)
            pred_feats = T(input_feats)
        frames = dec(pred_feats)
        return frames[:, :-num_pred], frames[:, -num_pred:]
    if mode not in ("RIP", "RIL"):
        raise ValueError("far_rollout: mode must be 'train', 'RIP' or 'RIL'")
    horizon = T.num_future_frames
    frames = [dec(pred_feats[:, -1:])]
    window = past_feats
    newest = pred_feats[:, -1:]                    # the first appended feature is the raw prediction in both modes
    for i in range(1, num_pred):
        window = torch.cat([window, newest], dim=1)
        if i > 1 and i >= horizon:
            window = window[:, 1:]                 # slide: drop the oldest feature once `horizon` predictions were appended
        pred_feats = T(window)
        frame = dec(pred_feats[:, -1:])
        frames.append(frame)
        newest = enc(frame) if mode == "RIP" else pred_feats[:, -1:]
    return torch.cat(frames, dim=1)
