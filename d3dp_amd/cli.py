"""`main.py`-compatible entrypoint: `--evaluate FILE` runs the hot path (reference main.py:596-794, README.md:39);
without it the training loop runs (reference main.py:305-593; d3dp_amd/trainer.py).

Keeps the reference's flag names for everything the path consumes (-c, --evaluate, -num_proposals,
-sampling_timesteps, -b, -f, -cs, -dep, -scale, -timestep, -gpu, --nolog, --debug) and its control flow:
flipped 2D copy (main.py:646-648), clip chunking (267-299), -b batching (682-698), root-joint zeroing (700),
trajectory add + reprojection (706-712), the four MPJPE aggregations (715-718), N-weighted accumulation (720-736)
and the `step %d : Protocol #1 Error (MPJPE) ...` lines written to <checkpoint>/h36m_test_log_H%d_K%d.txt (745-774).

`--p2` adds the Procrustes-aligned Protocol #2 lines (main.py:726-729, 776-783) through the batched device kernel.
Out of scope: the Human3.6M/3DHP dataset loaders and rendering.
`--synthetic` replaces main.py:83-145 with seeded synthetic sequences; without it the script explains what is missing.
Run under `python -m torch.distributed.run --nproc-per-node N main.py ...` to shard hypotheses over N GPUs.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

from . import D3DP, jpma
from .clips import clip_gather
from .dist import all_gather_hypotheses, hypothesis_slice, init_from_env, rank_generator
from .weights import H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, make_state_dict


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="D3DP evaluation on MI355X (libd3dp_hip)")
    p.add_argument('-d', '--dataset', default='h36m', type=str)
    p.add_argument('-k', '--keypoints', default='cpn_ft_h36m_dbb', type=str)
    p.add_argument('-c', '--checkpoint', default='checkpoint', type=str, help='checkpoint directory')
    p.add_argument('--evaluate', default='', type=str, metavar='FILENAME', help='checkpoint to evaluate (file name)')
    p.add_argument('--nolog', action='store_true')
    p.add_argument('-gpu', default='0', type=str)
    p.add_argument('-b', '--batch-size', default=4, type=int, help='clips per model call')
    p.add_argument('-cs', default=512, type=int)
    p.add_argument('-dep', default=8, type=int)
    p.add_argument('-f', '--number-of-frames', default=243, type=int)
    p.add_argument('-scale', default=1.0, type=float)
    p.add_argument('-timestep', type=int, default=1000)
    p.add_argument('-sampling_timesteps', type=int, default=5)
    p.add_argument('-num_proposals', type=int, default=300)
    p.add_argument('--debug', action='store_true', default=False)
    p.add_argument('--p2', action='store_true', default=False)
    # training (arguments.py:27-47, 68)
    p.add_argument('-cf', '--checkpoint-frequency', default=20, type=int)
    p.add_argument('-r', '--resume', default='', type=str, metavar='FILENAME')
    p.add_argument('-s', '--stride', default=243, type=int)
    p.add_argument('-e', '--epochs', default=400, type=int)
    p.add_argument('-lr', '--learning-rate', default=0.00006, type=float)
    p.add_argument('-lrd', '--lr-decay', default=0.993, type=float)
    p.add_argument('--coverlr', action='store_true')
    p.add_argument('-mloss', '--min_loss', default=100000, type=float)
    p.add_argument('-no-da', '--no-data-augmentation', dest='data_augmentation', action='store_false')
    p.add_argument('--no-eval', '--no_eval', dest='no_eval', action='store_true')
    # additions
    p.add_argument('--synthetic', action='store_true', help='seeded synthetic sequences instead of data/*.npz')
    p.add_argument('--synthetic-sequences', type=int, default=2)
    p.add_argument('--synthetic-frames', type=int, default=600)
    p.add_argument('--numerics', default=None, choices=['exact', 'fast'])
    p.add_argument('--seed', type=int, default=1)
    a = p.parse_args(argv)
    a.test_time_augmentation = True          # arguments.py:112 (no flag turns it off in the reference)
    return a


def synthetic_sequences(n_seq, n_frames, seed):
    """(cam(9), gt3d (N,17,3) camera space, kp2d (N,17,2) normalised) per sequence; 2D = projection of the 3D + noise."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cam = np.array([2.29, 2.287, 0.0254, 0.0289, -0.2070, 0.2477, -0.0030, -0.0009, -0.0014], np.float32)
    out = []
    for _ in range(n_seq):
        t = np.linspace(0, 6.0, n_frames, dtype=np.float32)[:, None, None]
        base = (rng.standard_normal((1, 17, 3)) * 0.25).astype(np.float32)
        sway = (rng.standard_normal((1, 17, 3)) * 0.05).astype(np.float32) * np.sin(t * (1 + rng.uniform(size=(1, 17, 3)).astype(np.float32)))
        pose = base + sway
        pose[:, 0] = 0
        traj = np.concatenate([0.3 * np.sin(t[:, 0]), 0.1 * np.cos(t[:, 0]), 4.0 + 0.2 * np.sin(0.5 * t[:, 0])], axis=-1)[:, None]
        absol = (pose + traj).astype(np.float32)
        kp = jpma.project_to_2d(torch.from_numpy(absol), torch.from_numpy(cam)).numpy()
        kp = kp + rng.normal(0, 0.005, kp.shape).astype(np.float32)
        gt = pose.astype(np.float32).copy()      # joints 1.. root-relative, joint 0 carries the trajectory
        gt[:, 0] = traj[:, 0]                    # (the layout main.py:99-103 gives the generator's 3D batches)
        out.append((cam, gt, kp.astype(np.float32)))
    return out


def load_model(args, device, h_local):
    model = D3DP(args, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, is_train=False, num_proposals=h_local,
                 sampling_timesteps=args.sampling_timesteps, numerics=args.numerics)
    path = os.path.join(args.checkpoint, args.evaluate) if args.evaluate else ''
    if path and os.path.exists(path):
        print('Loading evaluate checkpoint', path)
        # the reference's checkpoints carry a pickled numpy RandomState (main.py:546): not a weights-only file
        ck = torch.load(path, map_location='cpu', weights_only=False)
        sd = {k[len('module.'):] if k.startswith('module.') else k: v for k, v in ck['model_pos'].items()}
        model.load_state_dict(sd)
    elif args.synthetic:
        print('No checkpoint file: using seed-generated weights (d3dp_amd.weights, seed 7)')
        model.load_state_dict(make_state_dict(7, args.cs, args.dep, args.number_of_frames), strict=False)
    else:
        raise SystemExit(f"checkpoint {path!r} not found (pass --synthetic to run on seed-generated weights)")
    return model.to(device).eval()


def evaluate(args, model, sequences, device, rank, world, gen):
    K = args.sampling_timesteps
    names = ("J_Best", "P_Best", "P_Agg", "J_Agg")
    sums = {k: torch.zeros(K, device=device) for k in names}
    sums_p2 = {k: torch.zeros(K, device=device) for k in names}
    N = 0
    F_ = args.number_of_frames
    kl, kr = H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT
    with torch.no_grad():
        for cam, batch, batch_2d in sequences:
            s2 = torch.from_numpy(batch_2d.astype('float32')).to(device)
            s3 = torch.from_numpy(batch.astype('float32')).to(device)
            camt = torch.from_numpy(cam.astype('float32')).to(device)
            x2, x2f = clip_gather(s2, F_, kl, kr)          # main.py:646-648 (flip) + 267-299 (clips), one launch
            x3, _ = clip_gather(s3, F_)
            traj = x3[:, :, :1].clone()
            x3[:, :, 0] = 0
            bs = args.batch_size
            for i in range(0, x3.shape[0], bs):
                a2, a2f, a3, tr = x2[i:i + bs], x2f[i:i + bs], x3[i:i + bs], traj[i:i + bs]
                pred = model(a2.contiguous(), a3, input_2d_flip=a2f.contiguous(), generator=gen)   # (b,K,H_local,F,17,3)
                pred = all_gather_hypotheses(pred)                                                   # (b,K,H,F,17,3)
                pred[:, :, :, :, 0] = 0                                                              # main.py:700
                rp = jpma.reproject(pred, tr, camt)
                m = jpma.jpma_metrics(pred, a3, rp, a2)
                # the fused HIP kernel gives the aggregated poses (what a serving caller consumes) and the same J_Agg
                _, _, es, _ = jpma.jpma_hip(pred, tr, camt, a2, a3, zero_root=False, want_errors=True)
                m["J_Agg"] = es.permute(1, 0, 2, 3).reshape(K, -1).mean(-1)
                w = a3.shape[0] * a3.shape[1]
                for k in sums:
                    sums[k] += w * m[k]
                if args.p2:                                                                          # main.py:724-729
                    m2 = jpma.p_mpjpe_metrics(pred, a3, rp, a2)
                    for k in sums_p2:
                        sums_p2[k] += w * m2[k]
                N += w
                if args.debug:
                    break
            if args.debug:
                break
    out = {k: (v / N) * 1000 for k, v in sums.items()}
    out_p2 = {k: (v / N) * 1000 for k, v in sums_p2.items()} if args.p2 else None
    return out, out_p2, N


def run_evaluate(args, rank, world, device):
    sl = hypothesis_slice(args.num_proposals, rank, world)
    model = load_model(args, device, sl.stop - sl.start)
    seqs = synthetic_sequences(args.synthetic_sequences, args.synthetic_frames, args.seed)
    errs, errs_p2, N = evaluate(args, model, seqs, device, rank, world, rank_generator(args.seed, rank, device))
    if rank == 0:
        os.makedirs(args.checkpoint, exist_ok=True)
        log_path = os.path.join(args.checkpoint, 'h36m_test_log_H%d_K%d.txt' % (args.num_proposals, args.sampling_timesteps))
        with open(log_path, mode='a') as f:
            print('----synthetic----')
            f.write('----synthetic----\n')
            print('Test time augmentation:', True)
            for ii in range(args.sampling_timesteps):
                for name in ("J_Best", "P_Best", "P_Agg", "J_Agg"):
                    print('step %d : Protocol #1 Error (MPJPE) %s:' % (ii, name), errs[name][ii].item(), 'mm')
                    f.write('step %d : Protocol #1 Error (MPJPE) %s: %f mm\n' % (ii, name, errs[name][ii].item()))
            if errs_p2 is not None:                                                                  # main.py:776-783
                for ii in range(args.sampling_timesteps):
                    for name in ("J_Best", "P_Best", "P_Agg", "J_Agg"):
                        print('step %d : Protocol #2 Error (MPJPE) %s:' % (ii, name), errs_p2[name][ii].item(), 'mm')
                        f.write('step %d : Protocol #2 Error (MPJPE) %s: %f mm\n' % (ii, name, errs_p2[name][ii].item()))
            print('----------')
            f.write('----------\n')
        print(f'evaluated {N} frames; log appended to {log_path}')
    return 0


def run_train(args, rank, world, device):
    """main.py:305-593 on synthetic sequences: ChunkedBatcher (pools in HBM) -> D3DP(is_train=True) -> HipAdamW."""
    from .data import ChunkedBatcher
    from .trainer import fit
    if world > 1:
        raise SystemExit("training is single-GPU in this build (the reference's DataParallel replica split is not "
                         "reproduced, SURVEY.md §8 E1); launch without torch.distributed.run")
    kl, kr = H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT
    train = synthetic_sequences(args.synthetic_sequences, args.synthetic_frames, args.seed)
    valid = synthetic_sequences(max(1, args.synthetic_sequences // 2), args.synthetic_frames, args.seed + 1000)
    model_train = D3DP(args, kl, kr, is_train=True).to(device)
    model_eval = D3DP(args, kl, kr, is_train=False, numerics=args.numerics).to(device)
    if not args.resume:
        model_train.load_state_dict(make_state_dict(7, args.cs, args.dep, args.number_of_frames), strict=False)
    n_params = sum(p.numel() for p in model_train.parameters())
    print('INFO: Trainable parameter count:', n_params / 1000000, 'Million')
    batcher = ChunkedBatcher(max(1, args.batch_size // args.stride), [c for c, _, _ in train], [g for _, g, _ in train],
                             [k for _, _, k in train], args.number_of_frames, shuffle=True,
                             augment=args.data_augmentation, kps_left=kl, kps_right=kr, joints_left=kl, joints_right=kr,
                             device=device)
    print('INFO: Training on {} frames'.format(sum(g.shape[0] for _, g, _ in train)))
    hist = fit(args, model_train, model_eval, batcher, (lambda: valid) if not args.no_eval else None, device, kl, kr)
    return 0 if hist["losses_3d_train"] else 1


def main(argv=None):
    args = parse_args(argv)
    rank, world, local = init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("main.py needs an MI355X: libd3dp_hip has no CPU fallback")
    if not args.synthetic:
        raise SystemExit("dataset loading (data/data_3d_h36m.npz, data_2d_*.npz) is out of scope for this build "
                         "(SURVEY.md §2 rows 7-8); run with --synthetic")
    torch.cuda.set_device(local if world > 1 else int(args.gpu.split(',')[0]))
    device = torch.device('cuda', torch.cuda.current_device())
    if args.evaluate:
        return run_evaluate(args, rank, world, device)
    return run_train(args, rank, world, device)        # like the reference: no --evaluate means train (main.py:305)


if __name__ == '__main__':
    sys.exit(main())


# ---- MPI-INF-3DHP entrypoint (reference main_3dhp.py --evaluate) -------------------------------------------------------
def synthetic_sequences_3dhp(n_seq, n_frames, seed):
    """{key: (gt3d_mm (N,17,3) camera space with the root trajectory at joint 14, kp2d (N,17,2) normalised, valid (N,))}
    for keys TS1.. (TS5/TS6 use the second camera like the reference)."""
    from . import eval3dhp as e3
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for s in range(n_seq):
        key = "TS%d" % (s + 1)
        cam, data, linear = e3.camera_for(key)
        t = np.linspace(0, 6.0, n_frames, dtype=np.float32)[:, None, None]
        pose = (rng.standard_normal((1, 17, 3)) * 250).astype(np.float32) + \
               (rng.standard_normal((1, 17, 3)) * 50).astype(np.float32) * np.sin(t * (1 + rng.uniform(size=(1, 17, 3)).astype(np.float32)))
        pose[:, e3.ROOT_3DHP] = 0
        traj = np.concatenate([300 * np.sin(t[:, 0]), 100 * np.cos(t[:, 0]), 4000 + 200 * np.sin(0.5 * t[:, 0])], axis=-1)[:, None]
        absol = torch.from_numpy((pose + traj).astype(np.float32))
        XX = torch.clamp(absol[..., :2] / absol[..., 2:], -1, 1)
        pix = cam[:2] * XX + cam[2:4]                                             # camera.py:62-83
        w, h = float(data[0]), float(data[1])
        kp = (pix / w * 2 - torch.tensor([1.0, h / w])).numpy()                   # camera.py:7-11
        kp = kp + rng.normal(0, 0.003, kp.shape).astype(np.float32)
        gt = pose.astype(np.float32).copy()
        gt[:, e3.ROOT_3DHP] = traj[:, 0]
        out[key] = (gt, kp.astype(np.float32), (rng.uniform(size=n_frames) < 0.9).astype(np.float32))
    return out


def main_3dhp(argv=None):
    """`python main_3dhp.py --synthetic -c DIR --evaluate FILE -num_proposals H -sampling_timesteps K` : the reference's
    3DHP evaluation (main_3dhp.py:659-912) with D3DP3DHP, per-sequence P_Best / P_Agg errors in
    `3dhp_test_log_H%d_K%d.txt` and the four inference_data_*.mat pose files."""
    from . import D3DP3DHP, eval3dhp as e3
    args = parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("main_3dhp.py needs an MI355X: libd3dp_hip has no CPU fallback")
    if not args.synthetic or not args.evaluate:
        raise SystemExit("this build runs the 3DHP evaluation on --synthetic sequences only (dataset loaders and the 3DHP "
                         "training script are out of scope, SURVEY.md §2); pass --synthetic --evaluate FILE")
    torch.cuda.set_device(int(args.gpu.split(',')[0]))
    device = torch.device('cuda', torch.cuda.current_device())
    model = D3DP3DHP(args, e3.KPS_LEFT_3DHP, e3.KPS_RIGHT_3DHP, is_train=False, num_proposals=args.num_proposals,
                     sampling_timesteps=args.sampling_timesteps, numerics=args.numerics)
    path = os.path.join(args.checkpoint, args.evaluate)
    if os.path.exists(path):
        ck = torch.load(path, map_location='cpu', weights_only=False)
        model.load_state_dict({k[len('module.'):] if k.startswith('module.') else k: v for k, v in ck['model_pos'].items()})
    else:
        print('No checkpoint file: using seed-generated weights (d3dp_amd.weights, seed 7)')
        model.load_state_dict(make_state_dict(7, args.cs, args.dep, args.number_of_frames), strict=False)
    model = model.to(device).eval()
    os.makedirs(args.checkpoint, exist_ok=True)
    log_path = os.path.join(args.checkpoint, '3dhp_test_log_H%d_K%d.txt' % (args.num_proposals, args.sampling_timesteps))
    gen = rank_generator(args.seed, 0, device)
    stitched = {}
    for key, (gt, kp, valid) in synthetic_sequences_3dhp(args.synthetic_sequences, args.synthetic_frames, args.seed).items():
        sums, N, st = e3.evaluate_sequence(model, gt, kp, valid, key, args.number_of_frames, batch_clips=2, generator=gen)
        stitched[key] = st
        e1, e1_mean = sums["P_Best"] / N, sums["P_Agg"] / N
        with open(log_path, mode='a') as f:
            print('----' + key + '----')
            f.write('----' + key + '----\n')
            for ii in range(e1.shape[0]):
                print('step %d : Protocol #1 Error (MPJPE) P_Best:' % ii, e1[ii].item(), 'mm')
                f.write('step %d : Protocol #1 Error (MPJPE) P_Best: %f mm\n' % (ii, e1[ii].item()))
                print('step %d : Protocol #1 Error (MPJPE) P_Agg:' % ii, e1_mean[ii].item(), 'mm')
                f.write('step %d : Protocol #1 Error (MPJPE) P_Agg: %f mm\n' % (ii, e1_mean[ii].item()))
            print('----------')
            f.write('----------\n')
        if args.debug:
            break
    paths = e3.export_mat(args.checkpoint, stitched)
    print('wrote', ', '.join(sorted(paths.values())))
    return 0
