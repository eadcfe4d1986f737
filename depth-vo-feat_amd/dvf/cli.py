"""Shared plumbing of the entry scripts (train.py / unsupervise.py / unsupervise_dvo.py): process-group setup,
synthetic data stream, the step loop with device-side loss accumulation, checkpoints in the reference's format.

Data: the seeded synthetic stream that the benchmark uses by default; with ``--data-root`` the stereo-sequence listing
of the reference's un_dataset.py is read through this build's PIL-based ``un_dataset.dataset``."""
import os
import time

import torch
import torch.distributed as dist

from .engine import FlatAdam, GraphedStep
from .synthetic import synthetic_batch


def add_common_flags(parser):
    g = parser.add_argument_group("MI355X build extensions")
    g.add_argument("--synthetic", action="store_true", default=True,
                   help="seeded synthetic KITTI-shaped data (used when no --data-root listing is given)")
    g.add_argument("--data-root", default=None,
                   help="directory laid out like the reference's data/kitti_eigen (train.txt, intrinsics/, train_K/, "
                        "train_T_R2L/): read through un_dataset.dataset; each rank takes every world-th sample")
    g.add_argument("--height", type=int, default=256)
    g.add_argument("--width", type=int, default=832)
    g.add_argument("--steps-per-epoch", type=int, default=50, help="synthetic iterations per epoch")
    g.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    g.add_argument("--deterministic", action="store_true",
                   help="weight / bias gradients summed in a fixed order (no float atomics): runs from identical state are "
                        "bit-identical, as on the reference's CPU path; costs ~5 %% of a step")
    g.add_argument("--reference-stereo-pose", action="store_true",
                   help="feed the dataset's T_R2L file vector (0,0,0,Tx,0,0) unchanged into pose_vec2mat / the stereo-pose "
                        "MSE target, exactly as the reference does (unsupervise.py:101, train.py:201); default: convert it "
                        "to the (t, r-euler) order those functions read (un_dataset.to_batch)")


def init_distributed():
    """One process per GPU; RANK/LOCAL_RANK/WORLD_SIZE from torchrun.  Returns (rank, world, device)."""
    if not torch.cuda.is_available():
        raise SystemExit("this build runs on MI355X only: no CUDA/HIP device visible and there is no CPU fallback")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    return rank, world, device


def load_pretrained(module, path, strict=True):
    """Reference checkpoint convention (train.py:129-141): a dict with key 'state_dict', or a bare state_dict."""
    weights = torch.load(path, map_location="cpu", weights_only=True)
    module.load_state_dict(weights["state_dict"] if "state_dict" in weights else weights, strict=strict)


def save_best(output_dir, named_modules):
    """Reference file names (train.py:242-247, unsupervise.py:156-163): bare state_dicts."""
    os.makedirs(output_dir, exist_ok=True)
    for fname, module in named_modules:
        # parameters are views into FlatAdam's single arena: clone, or torch.save would serialise the whole arena
        # (every network's weights) into each file
        torch.save({k: v.detach().cpu().clone() for k, v in module.state_dict().items()}, os.path.join(output_dir, fname))


@torch.no_grad()
def validate(pose_net, val_loader, device):
    """Reference train.py:220-247: mean over the validation SAMPLES of the per-batch L1 between the predicted relative
    pose (network output 0 through the se(3) exponential map, se3_generate.py) and the ground-truth 4x4 transform.
    The pose network runs in eval mode on (later frame, [earlier frame, earlier frame])."""
    from se3_generate import generate_se3
    was_training = pose_net.training
    pose_net.eval()
    total = torch.zeros((), device=device)
    for data, target in val_loader:
        data, target = data.float().to(device), target.float().to(device)
        tgt, ref = data[:, :3].contiguous(), data[:, 3:].contiguous()
        _, pose = pose_net(tgt, [ref, ref])
        out = generate_se3(pose[:, 0].reshape(-1, 6, 1, 1)).view(-1, 4, 4)
        total += torch.nn.functional.l1_loss(out, target)                 # (the reference sums the per-batch means, :232)
    pose_net.train(was_training)
    return float(total) / max(len(val_loader.dataset), 1)


def run_training(args, nets, loss_fn, lr, betas, weight_decay, term_names, ckpt_names, n_views=2, val_fn=None):
    """Epoch loop.  ``loss_fn(batch) -> (loss, terms)``; ``nets``: list of modules in optimizer-group order."""
    rank, world, device = args._rank, args._world, args._device
    if getattr(args, "deterministic", False):
        from . import conv as _conv
        _conv.set_deterministic(True)
    params = [p for net in nets for p in net.parameters()]
    opt = FlatAdam(params, lr=lr, betas=betas, weight_decay=weight_decay, world_size=world, overlap=True)
    batch = synthetic_batch(args.batch_size, args.height, args.width, seed=1234 + args.seed, rank=rank, n_views=n_views,
                            device=device)
    # real data: un_dataset.dataset -> DataLoader (sharded by rank); every step copies the next batch INTO the static
    # input tensors, so the step (eager or graph replay) always reads the same addresses
    feed = None
    if getattr(args, "data_root", None):
        import un_dataset
        ds = un_dataset.dataset(img_height=args.height, img_width=args.width, root=args.data_root, seed=args.seed)
        idx = list(range(rank, len(ds), world))
        loader = torch.utils.data.DataLoader(torch.utils.data.Subset(ds, idx), batch_size=args.batch_size, shuffle=True,
                                             num_workers=getattr(args, "workers", 0), drop_last=True, pin_memory=True)
        args.steps_per_epoch = max(1, len(loader))

        def feed():
            while True:
                for sample in loader:
                    yield un_dataset.to_batch(sample, device, getattr(args, "reference_stereo_pose", False))
        feed = feed()
    KEYS = ("img_R2", "img_R1", "img_L2", "K", "Kinv", "T_R2L", "T_R2L_se3")

    def load_next():
        nxt = next(feed)
        for k in KEYS:
            batch[k].copy_(nxt[k].reshape(batch[k].shape), non_blocking=True)

    if feed is not None:
        load_next()              # the warm-up / capture steps below already train: they must see real data, not noise
    acc = torch.zeros(len(term_names), device=device)            # device-side running sums: no per-step .item()

    def step():
        loss, terms = loss_fn(batch)
        opt.zero_grad()
        loss.backward()
        opt.step()
        acc.add_(torch.stack([terms[k] for k in term_names]))
        return (terms["total"],)

    runner = step
    if world == 1 and not args.no_graph:
        step()                                                   # one eager step sizes every workspace
        runner = GraphedStep(step, [], warmup=1)
    best = float("inf")
    for epoch in range(args.epochs):
        acc.zero_()
        t0 = time.perf_counter()
        for it in range(args.steps_per_epoch):
            if feed is not None and (epoch, it) != (0, 0):
                load_next()
            runner()
            if getattr(args, "log_interval", 0) and rank == 0 and (it + 1) % args.log_interval == 0:
                # (one host sync per log interval; the reference syncs on five .item() calls every step, train.py:205-209)
                print(f"  epoch {epoch} [{it + 1}/{args.steps_per_epoch}] {term_names[0]}: "
                      f"{float(acc[0]) / (it + 1):.6f}", flush=True)
            # a fresh synthetic batch every step would only change values, not the work; new data is copied into the
            # static input tensors in place (graph replay reads the same addresses)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        means = (acc / args.steps_per_epoch).tolist()
        if world > 1:
            tt = torch.tensor(means, device=device)
            dist.all_reduce(tt)
            means = (tt / world).tolist()
        if rank == 0:
            msg = " ".join(f"{k}: {v:.9f}" for k, v in zip(term_names, means))
            print(f"Train epoch {epoch}: {msg}  [{args.batch_size * world * args.steps_per_epoch / dt:.1f} samples/s]", flush=True)
            # checkpoint criterion: the reference's validate() (train.py:238-247) when a validation set is given,
            # else the best training loss
            score = means[0]
            if val_fn is not None:
                score = val_fn()
                print("Test set: Average loss: {:.6f} [BEST:{}]".format(score, score < best), flush=True)
            if score < best:
                best = score
                save_best(args.output_dir, ckpt_names)
    if world > 1:
        dist.destroy_process_group()
