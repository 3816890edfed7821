"""Test-time rollouts of VPTR on the MI355X path.

`nar_rollout`   -- Test_VPTR.ipynb cell 5: one NAR pass predicts Tf frames; longer horizons chain passes, each taking the last
                   Tp predicted frames (re-encoded) as the new past.
`far_rollout`   -- train_FAR.py:103-125 (test_phase=True): the FAR transformer predicts one feature per step; from the second
                   step on the newest prediction is decoded to a frame and re-encoded (Dec -> Enc) before it is appended.
Everything runs under no_grad through the same HIP kernels as training (eval mode: dropout / DropPath off).
"""
import torch


@torch.no_grad()
def nar_rollout(enc, dec, T, past, rounds=1):
    """past (N,Tp,C,H,W) -> predicted frames (N, rounds*Tf, C, H, W)."""
    T.eval()
    Tp = past.shape[1]
    out = []
    cur = past
    for _ in range(rounds):
        pred = dec(T(enc(cur)))
        out.append(pred)
        hist = torch.cat([cur, pred], dim=1)
        cur = hist[:, -Tp:]
    return torch.cat(out, dim=1)


@torch.no_grad()
def far_rollout(enc, dec, T, past, num_pred):
    """past (N,Tp,C,H,W) -> (pred_past_frames (N,Tp-1,...), pred_future_frames (N,num_pred,...)) exactly as
    FAR_show_sample's test phase (train_FAR.py:112-128)."""
    T.eval()
    past_feats = enc(past)
    pred_feats = T(past_feats)
    input_feats = past_feats
    for i in range(num_pred - 1):
        if i == 0:
            input_feats = torch.cat([past_feats, pred_feats[:, -1:]], dim=1)
        else:
            frame = dec(pred_feats[:, -1:])
            input_feats = torch.cat([input_feats, enc(frame)], dim=1)
        pred_feats = T(input_feats)
    frames = dec(pred_feats)
    return frames[:, :-num_pred], frames[:, -num_pred:]
