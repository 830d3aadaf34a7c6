"""Per-iteration LR: linear warm-up then half-cosine (reference util/lr_sched.py:9-27), honours `lr_scale`."""
import math


def adjust_learning_rate(optimizer, epoch, args):
    if epoch < args.warmup_epochs:
        lr = args.lr * epoch / args.warmup_epochs
    else:
        progress = (epoch - args.warmup_epochs) / (args.epochs - args.warmup_epochs)
        lr = args.min_lr + (args.lr - args.min_lr) * 0.5 * (1.0 + math.cos(math.pi * progress))
    for group in optimizer.param_groups:
        group["lr"] = lr * group["lr_scale"] if "lr_scale" in group else lr
    return lr
