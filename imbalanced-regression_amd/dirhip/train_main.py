"""``train.py`` driver — the command line of ``imdb-wiki-dir/train.py`` / ``agedb-dir/train.py`` (every flag and
default of ``train.py:23-72``), the same store-name / checkpoint layout (``:78-93``, ``:209-215``), the same epoch
loop, ``validate`` and ``shot_metrics`` — run as one process per MI355X.

Differences from the reference, all on purpose:
  * nothing happens at import time (argparse / folder creation / log files live inside ``run()``), so the
    module is usable as a library and from tests;
  * ``torch.nn.DataParallel`` (``train.py:143``) -> ``dirhip.parallel.DataParallelEngine`` under ``torchrun``
    (``python -m torch.distributed.run --nproc-per-node N train.py ...``); with one process it is a plain wrapper;
  * the hot loop and the FDS epoch tail come from ``dirhip.train_loop`` (device-resident features, loss read
    back every ``print_freq`` steps instead of every step);
  * extra, optional flags: ``--synthetic N`` (random images, no files needed), ``--amp {bf16,fp32,fp32x3,fp32x2}``, ``--amp_early``,
    ``--max_steps`` (truncate epochs for smoke runs). ``tensorboard_logger`` is used when importable.
"""
import argparse
import logging
import os
import time
from collections import OrderedDict, defaultdict

import numpy as np
import torch
import torch.nn as nn
from scipy.stats import gmean

from .parallel import DataParallelEngine, init_distributed, shard_indices
from .resnet import resnet50
from .train_loop import EpochFeatures, epoch_tail, resolve_loss, train_step
from .optim import SGD, Adam
from .utils import AverageMeter, ProgressMeter, adjust_learning_rate, prepare_folders, save_checkpoint

print = logging.info

DATASET_DEFAULTS = {                       # the five argparse defaults that differ (train.py:29,35,40,50)
    'imdb_wiki': dict(lds_ks=5, fds_ks=5, bucket_start=0),
    'agedb': dict(lds_ks=9, fds_ks=9, bucket_start=3),
}


def build_parser(dataset_default='imdb_wiki'):
    d = DATASET_DEFAULTS[dataset_default]
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    # LDS
    p.add_argument('--lds', action='store_true', default=False, help='whether to enable LDS')
    p.add_argument('--lds_kernel', type=str, default='gaussian', choices=['gaussian', 'triang', 'laplace'], help='LDS kernel type')
    p.add_argument('--lds_ks', type=int, default=d['lds_ks'], help='LDS kernel size: should be odd number')
    p.add_argument('--lds_sigma', type=float, default=1, help='LDS gaussian/laplace kernel sigma')
    # FDS
    p.add_argument('--fds', action='store_true', default=False, help='whether to enable FDS')
    p.add_argument('--fds_kernel', type=str, default='gaussian', choices=['gaussian', 'triang', 'laplace'], help='FDS kernel type')
    p.add_argument('--fds_ks', type=int, default=d['fds_ks'], help='FDS kernel size: should be odd number')
    p.add_argument('--fds_sigma', type=float, default=1, help='FDS gaussian/laplace kernel sigma')
    p.add_argument('--start_update', type=int, default=0, help='which epoch to start FDS updating')
    p.add_argument('--start_smooth', type=int, default=1, help='which epoch to start using FDS to smooth features')
    p.add_argument('--bucket_num', type=int, default=100, help='maximum bucket considered for FDS')
    p.add_argument('--bucket_start', type=int, default=d['bucket_start'], choices=[0, 3],
                   help='minimum(starting) bucket for FDS, 0 for IMDBWIKI, 3 for AgeDB')
    p.add_argument('--fds_mmt', type=float, default=0.9, help='FDS momentum')
    # re-weighting / two-stage
    p.add_argument('--reweight', type=str, default='none', choices=['none', 'sqrt_inv', 'inverse'], help='cost-sensitive reweighting scheme')
    p.add_argument('--retrain_fc', action='store_true', default=False, help='whether to retrain last regression layer (regressor)')
    # training / optimisation
    p.add_argument('--dataset', type=str, default=dataset_default, choices=['imdb_wiki', 'agedb'], help='dataset name')
    p.add_argument('--data_dir', type=str, default='./data', help='data directory')
    p.add_argument('--model', type=str, default='resnet50', help='model name')
    p.add_argument('--store_root', type=str, default='checkpoint', help='root path for storing checkpoints, logs')
    p.add_argument('--store_name', type=str, default='', help='experiment store name')
    p.add_argument('--gpu', type=int, default=None)
    p.add_argument('--optimizer', type=str, default='adam', choices=['adam', 'sgd'], help='optimizer type')
    p.add_argument('--loss', type=str, default='l1', choices=['mse', 'l1', 'focal_l1', 'focal_mse', 'huber'], help='training loss type')
    p.add_argument('--lr', type=float, default=1e-3, help='initial learning rate')
    p.add_argument('--epoch', type=int, default=90, help='number of epochs to train')
    p.add_argument('--momentum', type=float, default=0.9, help='optimizer momentum')
    p.add_argument('--weight_decay', type=float, default=1e-4, help='optimizer weight decay')
    p.add_argument('--schedule', type=int, nargs='*', default=[60, 80], help='lr schedule (when to drop lr by 10x)')
    p.add_argument('--batch_size', type=int, default=256, help='batch size (per process)')
    p.add_argument('--print_freq', type=int, default=10, help='logging frequency')
    p.add_argument('--img_size', type=int, default=224, help='image size used in training')
    p.add_argument('--workers', type=int, default=32, help='number of workers used in data loading')
    # checkpoints
    p.add_argument('--resume', type=str, default='', help='checkpoint file path to resume training')
    p.add_argument('--pretrained', type=str, default='', help='checkpoint file path to load backbone weights')
    p.add_argument('--evaluate', action='store_true', help='evaluate only flag')
    # additions
    p.add_argument('--synthetic', type=int, default=0, help='train on N synthetic samples (no image files)')
    p.add_argument('--amp', type=str, default='bf16', choices=['bf16', 'fp32', 'fp32x3', 'fp32x2'], help='conv-stack precision: bf16 MFMA (default); fp32 = exact '
                   'float32 MFMA (the reference\'s arithmetic, the parity mode); fp32x3 / fp32x2 = float32 tensors on the bf16 matrix pipe with every operand split '
                   'into three / two bf16 terms: float32-grade results (x3: at float32\'s own rounding noise; x2: 16 significand bits), 1.4x / 2x the speed of fp32')
    p.add_argument('--amp_switch_epoch', type=int, default=None, help='precision schedule: epochs < E run the conv stack in float32 (--amp_early), epochs >= E '
                   'in bf16 (--amp bf16). Measured on the val-MAE proxy: bf16 costs MAE only during the first part of a schedule (random initialisation, full '
                   'learning rate); from a float32 start it trains to the same MAE')
    p.add_argument('--amp_early', type=str, default='fp32', choices=['fp32', 'fp32x3', 'fp32x2'], help='arithmetic of the epochs before --amp_switch_epoch')
    p.add_argument('--max_steps', type=int, default=0, help='truncate every epoch to this many steps (0 = full)')
    p.add_argument('--gpu_augment', action='store_true', help='image files only: the DataLoader yields decoded, resized uint8 images and '
                   'RandomCrop / flip / ToTensor / Normalize run as one HIP kernel per batch (dir_augment_u8) instead of per image on the host')
    p.add_argument('--gpu_resize', action='store_true', help='image files only, implies --gpu_augment: loader workers only DECODE; Resize((S, S)) runs on the '
                   'GPU as well (dir_resize_u8: Pillow\'s bilinear arithmetic bit for bit, on the ragged uint8 batch), then dir_augment_u8')
    p.add_argument('--gpu_cache', action='store_true', help='image files only, with --gpu_augment / --gpu_resize: keep the resized uint8 training images in HBM '
                   '(datasets.DeviceImageCache: 28.8 GB for IMDB-WIKI at 224); the FDS feature pass of the same epoch and every later epoch gather them '
                   'there and draw a fresh augmentation on the GPU: no JPEG decode after the first training pass. NOTE (more than one rank): the partition '
                   'of the training set over the ranks is then FIXED for the whole run (every rank caches its own shard; batches never mix samples of '
                   'different shards and a rank\'s BatchNorm statistics always see the same subset) — unlike the default path, which redraws the partition '
                   'every epoch like a DistributedSampler; every rank allocates the full-size cache')
    p.add_argument('--overwrite', action='store_true', help='delete an existing run folder of the same name (the reference asks on '
                   'the terminal; without a terminal nothing is deleted unless this flag is given)')
    p.set_defaults(augment=True)
    return p


def make_store_name(args):
    """train.py:78-93."""
    name = f'_{args.store_name}' if len(args.store_name) else ''
    if not args.lds and args.reweight != 'none':
        name += f'_{args.reweight}'
    if args.lds:
        name += f'_lds_{args.lds_kernel[:3]}_{args.lds_ks}'
        if args.lds_kernel in ['gaussian', 'laplace']:
            name += f'_{args.lds_sigma}'
    if args.fds:
        name += f'_fds_{args.fds_kernel[:3]}_{args.fds_ks}'
        if args.fds_kernel in ['gaussian', 'laplace']:
            name += f'_{args.fds_sigma}'
        name += f'_{args.start_update}_{args.start_smooth}_{args.fds_mmt}'
    if args.retrain_fc:
        name += '_retrain_fc'
    return f"{args.dataset}_{args.model}{name}_{args.optimizer}_{args.loss}_{args.lr}_{args.batch_size}"


class _NullTB:
    def log_value(self, *a, **k):
        pass


def _loader_batches(loader, device, augment=None, resize=None, cache=None):
    """``augment``: a ``datasets.DeviceAugment`` when the dataset is in raw mode — the uint8 batch goes over PCIe (a quarter of the
    float32 bytes) and crop / flip / normalise / cast run as one kernel on the GPU. ``resize``: a ``datasets.DeviceResize``
    when the workers only decode (``raw="decoded"`` + ``ragged_collate``): the batch arrives as (flat bytes, sizes, ...). On a GPU the
    device half runs ahead of the loop in ``datasets.DevicePrefetcher`` (side stream + thread): the copies never wait for the training stream.
    ``cache``: a ``datasets.DeviceImageCache``; the batch then carries its sample indices as the last item and its resized bytes are stored."""
    def to_device(batch):
        index = None
        if cache is not None:
            index, batch = batch[-1], tuple(batch[:-1])
        if resize is not None:
            inputs = resize(batch[0], batch[1])
            batch = (inputs,) + tuple(batch[2:])
        inputs, targets, weights = batch[0], batch[1], batch[2]
        inputs = inputs.to(device, non_blocking=True)
        targets, weights = targets.to(device, non_blocking=True), weights.to(device, non_blocking=True)
        if cache is not None:
            cache.put(index, inputs, targets, weights)
        if augment is not None:
            inputs = augment(inputs)
        out = (inputs, targets, weights)
        return out + ((batch[3].bool(),) if len(batch) > 3 else ())
    if torch.device(device).type == "cuda":
        from .datasets import DevicePrefetcher
        pf = DevicePrefetcher(loader, device, to_device)
        try:
            yield from pf
        finally:
            pf.close()
    else:
        for batch in loader:
            yield to_device(batch)


class _ShardSubset(torch.utils.data.Dataset):
    """This rank's shard of a dataset; every item also carries whether it is a real sample of the epoch or one of the
    wrap-around duplicates that pad the shard (``parallel.shard_indices``)."""

    def __init__(self, dataset, indices, valid, with_index=False):
        self.dataset, self.indices, self.valid, self.with_index = dataset, indices, valid, with_index

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, i):
        out = tuple(self.dataset[self.indices[i]]) + (bool(self.valid[i]),)
        return out + ((int(self.indices[i]),) if self.with_index else ())           # (the dataset index: key of datasets.DeviceImageCache)


def _check_loss(v):
    assert not (np.isnan(v) or v > 1e6), f"Loss explosion: {v}"
    return v


def train(train_batches, n_steps, model, optimizer, epoch, args, store):
    """train.py:234-283. ``train_batches()`` returns a fresh iterator of device (inputs, targets, weights[, valid])."""
    batch_time = AverageMeter('Time', ':6.2f')
    losses = AverageMeter(f'Loss ({args.loss.upper()})', ':.3f')
    progress = ProgressMeter(n_steps, [batch_time, losses], prefix="Epoch: [{}]".format(epoch))
    loss_fn = resolve_loss(args.loss)
    model.train()
    end = time.time()
    pending = []
    for idx, batch in enumerate(train_batches()):
        inputs, targets, weights = batch[0], batch[1], batch[2]
        if args.max_steps and idx >= args.max_steps:
            break
        loss = train_step(model, optimizer, inputs, targets, weights, epoch, loss_fn, fds=args.fds)
        pending.append((loss, inputs.size(0)))
        if idx % args.print_freq == 0 or idx == n_steps - 1:
            for l, n in pending:                       # one host sync per print_freq steps (train.py:256-258 syncs every step)
                losses.update(_check_loss(l.item()), n)
            pending = []
            batch_time.update(time.time() - end)
            end = time.time()
            progress.display(idx)
    for l, n in pending:
        losses.update(_check_loss(l.item()), n)

    if args.fds and epoch >= args.start_update:
        print(f"Create Epoch [{epoch}] features of all training data...")

        def tail_batches():
            for i, batch in enumerate(train_batches()):
                if args.max_steps and i >= args.max_steps:
                    break
                yield (batch[0], batch[1]) + ((batch[3],) if len(batch) > 3 else ())
        epoch_tail(model, tail_batches(), epoch, store)
    return losses.avg


def validate(val_batches, n_batches, model, args, train_labels=None, prefix='Val'):
    """train.py:286-335. Same numbers as the reference: per-batch float32 MSE / L1 means averaged with the batch sizes as
    weights (``AverageMeter``), G-Mean of all absolute errors, per-shot metrics — but the per-batch values are read back
    once at the end instead of with two ``.item()`` host syncs per batch."""
    losses_mse = AverageMeter('Loss (MSE)', ':.3f')
    losses_l1 = AverageMeter('Loss (L1)', ':.3f')
    progress = ProgressMeter(n_batches, [losses_mse, losses_l1], prefix=f'{prefix}: ')
    model.eval()
    preds, labels, per_batch = [], [], []
    with torch.no_grad():
        for idx, (inputs, targets, _) in enumerate(val_batches()):
            outputs = model(inputs).float()
            preds.append(outputs)
            labels.append(targets)
            per_batch.append((nn.functional.mse_loss(outputs, targets), nn.functional.l1_loss(outputs, targets), inputs.size(0)))
    for i, (m, l, n) in enumerate(per_batch):
        losses_mse.update(m.item(), n)
        losses_l1.update(l.item(), n)
        if i % args.print_freq == 0:
            progress.display(i)
    preds = torch.cat(preds).cpu().numpy()
    labels = torch.cat(labels).cpu().numpy()
    err = np.abs(preds - labels)
    shot_dict = shot_metrics(np.hstack(preds), np.hstack(labels), train_labels)
    loss_gmean = gmean(np.hstack(err), axis=None).astype(float)
    print(f" * Overall: MSE {losses_mse.avg:.3f}\tL1 {losses_l1.avg:.3f}\tG-Mean {loss_gmean:.3f}")
    for shot, title in (('many', 'Many'), ('median', 'Median'), ('low', 'Low')):
        print(f" * {title}: MSE {shot_dict[shot]['mse']:.3f}\tL1 {shot_dict[shot]['l1']:.3f}\tG-Mean {shot_dict[shot]['gmean']:.3f}")
    return losses_mse.avg, losses_l1.avg, loss_gmean


def shot_metrics(preds, labels, train_labels, many_shot_thr=100, low_shot_thr=20):
    """train.py:338-391: per-label error sums split by the label's train-set frequency
    (many > 100, low < 20, median otherwise)."""
    train_labels = np.array(train_labels).astype(int)
    if isinstance(preds, torch.Tensor):
        preds, labels = preds.detach().cpu().numpy(), labels.detach().cpu().numpy()
    elif not isinstance(preds, np.ndarray):
        raise TypeError(f'Type ({type(preds)}) of predictions not supported')
    acc = {s: dict(mse=0.0, l1=0.0, cnt=0, all=[]) for s in ('many', 'median', 'low')}
    for l in np.unique(labels):
        n_train = int(np.sum(train_labels == l))
        sel = labels == l
        e = np.abs(preds[sel] - labels[sel])
        shot = 'many' if n_train > many_shot_thr else ('low' if n_train < low_shot_thr else 'median')
        acc[shot]['mse'] += float(np.sum(e ** 2))
        acc[shot]['l1'] += float(np.sum(e))
        acc[shot]['cnt'] += int(sel.sum())
        acc[shot]['all'].append(e)
    out = defaultdict(dict)
    for shot, a in acc.items():
        with np.errstate(all='ignore'):
            out[shot]['mse'] = a['mse'] / a['cnt'] if a['cnt'] else float('nan')
            out[shot]['l1'] = a['l1'] / a['cnt'] if a['cnt'] else float('nan')
            out[shot]['gmean'] = gmean(np.hstack(a['all']), axis=None).astype(float) if a['all'] else float('nan')
    return out


def run(argv=None, dataset_default='imdb_wiki'):
    args, _unknown = build_parser(dataset_default).parse_known_args(argv)
    args.start_epoch, args.best_loss = 0, 1e5
    args.store_name = make_store_name(args)
    rank, world, local_rank = init_distributed()
    if not torch.cuda.is_available():
        raise SystemExit("train.py needs an AMD GPU: the LDS/FDS hot path runs as HIP kernels only (no CPU fallback)")
    device = torch.device('cuda', local_rank if args.gpu is None else args.gpu)
    torch.cuda.set_device(device)

    if rank == 0:
        prepare_folders(args)
    if world > 1:
        torch.distributed.barrier()
    logging.root.handlers = []
    handlers = [logging.StreamHandler()]
    if rank == 0:
        handlers.append(logging.FileHandler(os.path.join(args.store_root, args.store_name, 'training.log')))
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.WARNING, format="%(asctime)s | %(message)s", handlers=handlers)
    print(f"Args: {args}")
    print(f"Store name: {args.store_name}")
    try:
        from tensorboard_logger import Logger
        tb_logger = Logger(logdir=os.path.join(args.store_root, args.store_name), flush_secs=2) if rank == 0 else _NullTB()
    except ImportError:
        tb_logger = _NullTB()

    # ---- data
    print('=====> Preparing data...')
    from . import datasets
    if args.synthetic:
        rng = np.random.default_rng(0)
        pool = np.clip(np.round(np.abs(rng.normal(0, 18, args.synthetic + 2000)) + 20), 0, 120).astype(np.float32)
        train_labels = pool[:args.synthetic]
        train_set = datasets.SyntheticAgeDataset(train_labels, args.img_size, args.reweight, args.lds, args.lds_kernel,
                                                 args.lds_ks, args.lds_sigma, seed=1)
        val_set = datasets.SyntheticAgeDataset(pool[args.synthetic:args.synthetic + 1000], args.img_size, seed=2)
        test_set = datasets.SyntheticAgeDataset(pool[args.synthetic + 1000:], args.img_size, seed=3)
        n_train = len(train_set)

        def train_batches(epoch):
            idx, valid = shard_indices(n_train, rank, world, epoch_seed=epoch, with_valid=True)
            return lambda: train_set.device_batches(idx, args.batch_size, device, seed=epoch, valid=valid)

        def eval_batches(ds):
            return lambda: ds.device_batches(torch.arange(len(ds)), args.batch_size, device)
        steps_per_epoch = (len(shard_indices(n_train, rank, world)) + args.batch_size - 1) // args.batch_size
        n_val = (len(val_set) + args.batch_size - 1) // args.batch_size
    else:
        import pandas as pd
        from torch.utils.data import DataLoader
        print(f"File (.csv): {args.dataset}.csv")
        df = pd.read_csv(os.path.join(args.data_dir, f"{args.dataset}.csv"))
        df_train, df_val, df_test = df[df['split'] == 'train'], df[df['split'] == 'val'], df[df['split'] == 'test']
        train_labels = df_train['age']
        cls = datasets.IMDBWIKI if args.dataset == 'imdb_wiki' else datasets.AgeDB
        raw = "decoded" if args.gpu_resize else bool(args.gpu_augment)
        train_set = cls(data_dir=args.data_dir, df=df_train, img_size=args.img_size, split='train', reweight=args.reweight,
                        lds=args.lds, lds_kernel=args.lds_kernel, lds_ks=args.lds_ks, lds_sigma=args.lds_sigma, raw=raw)
        val_set = cls(data_dir=args.data_dir, df=df_val, img_size=args.img_size, split='val', raw=raw)
        test_set = cls(data_dir=args.data_dir, df=df_test, img_size=args.img_size, split='test', raw=raw)
        n_train = len(train_set)
        aug_dtype = torch.bfloat16 if args.amp == 'bf16' else torch.float32
        aug_train = datasets.DeviceAugment(args.img_size, train=True, dtype=aug_dtype) if raw else None
        aug_eval = datasets.DeviceAugment(args.img_size, train=False, dtype=aug_dtype) if raw else None
        dev_resize = datasets.DeviceResize(args.img_size, device) if args.gpu_resize else None     # (a pinned staging ring measured slower: tools/pinned_stager.py)
        collate = datasets.ragged_collate if args.gpu_resize else None

        # --gpu_resize hands over RAGGED file-size batches (~4x the bytes, a different size every time): pageable, like the configuration
        # bench.py's input_pipeline leg measures — a pinned pool of varying block sizes takes tens of seconds to build and fragments the host
        # caching allocator; fixed-size batches stay pinned
        loader_kw = dict(num_workers=args.workers, pin_memory=not args.gpu_resize, collate_fn=collate)
        if args.gpu_resize and args.workers > 0:
            loader_kw["prefetch_factor"] = 2

        cache = None
        if args.gpu_cache:
            assert raw, "--gpu_cache keeps the uint8 images the GPU augmentation consumes: give --gpu_augment or --gpu_resize"
            cache = datasets.DeviceImageCache(n_train, args.img_size, device)
            print(f"Training images cached in HBM once decoded: {cache.u8.numel() / 2 ** 30:.2f} GiB")

        def train_batches(epoch):
            # (with the HBM cache under data parallelism the partition of the training set over the ranks is FIXED — every rank then finds its
            #  whole shard cached from the second pass on; the order inside the shard is still drawn anew for every pass. Without the cache the
            #  partition is redrawn every epoch, like a DistributedSampler's)
            idx, valid = shard_indices(n_train, rank, world, epoch_seed=0 if (cache is not None and world > 1) else epoch, with_valid=True)

            def batches():
                # (decided at the START of every pass: the feature pass of an epoch already finds what its training pass stored)
                if cache is not None and cache.covers(idx):
                    return cache.batches(idx, args.batch_size, aug_train, valid=valid)
                loader = DataLoader(_ShardSubset(train_set, idx.tolist(), valid.tolist(), with_index=cache is not None), batch_size=args.batch_size,
                                    shuffle=True, drop_last=False, **loader_kw)
                return _loader_batches(loader, device, aug_train, dev_resize, cache)
            return batches

        def eval_batches(ds):
            loader = DataLoader(ds, batch_size=args.batch_size, shuffle=False, **loader_kw)
            return lambda: _loader_batches(loader, device, aug_eval, dev_resize)
        steps_per_epoch = (len(shard_indices(n_train, rank, world)) + args.batch_size - 1) // args.batch_size
        n_val = (len(val_set) + args.batch_size - 1) // args.batch_size
    print(f"Training data size: {len(train_set)}")
    print(f"Validation data size: {len(val_set)}")
    print(f"Test data size: {len(test_set)}")

    # ---- model
    print('=====> Building model...')
    model = resnet50(fds=args.fds, bucket_num=args.bucket_num, bucket_start=args.bucket_start,
                     start_update=args.start_update, start_smooth=args.start_smooth,
                     kernel=args.fds_kernel, ks=args.fds_ks, sigma=args.fds_sigma, momentum=args.fds_mmt)
    fp32_first = args.amp != 'bf16' or (args.amp_switch_epoch is not None and args.start_epoch < args.amp_switch_epoch)
    arith_of = {'fp32': 'exact', 'fp32x3': 'x3', 'fp32x2': 'x2'}
    f32_arith = arith_of[args.amp] if args.amp != 'bf16' else arith_of[args.amp_early]
    model = DataParallelEngine(model.to(device), amp_dtype=None if fp32_first else torch.bfloat16, channels_last=True, f32_arith=f32_arith)

    if args.evaluate:
        assert args.resume, 'Specify a trained model using [args.resume]'
        checkpoint = torch.load(args.resume, map_location=device)
        model.load_state_dict(checkpoint['state_dict'], strict=False)
        print(f"===> Checkpoint '{args.resume}' loaded (epoch [{checkpoint['epoch']}]), testing...")
        n_test = (len(test_set) + args.batch_size - 1) // args.batch_size
        validate(eval_batches(test_set), n_test, model, args, train_labels=train_labels, prefix='Test')
        return

    if args.retrain_fc:
        assert args.reweight != 'none' and args.pretrained
        print('===> Retrain last regression layer only!')
        for name, param in model.named_parameters():
            if 'fc' not in name and 'linear' not in name:
                param.requires_grad = False

    parameters = list(filter(lambda p: p.requires_grad, model.parameters()))
    if args.retrain_fc:
        assert 1 <= len(parameters) <= 2  # fc.weight, fc.bias
    optimizer = Adam(parameters, lr=args.lr) if args.optimizer == 'adam' else \
        SGD(parameters, lr=args.lr, momentum=args.momentum, weight_decay=args.weight_decay)

    if args.pretrained:
        checkpoint = torch.load(args.pretrained, map_location="cpu")
        new_state_dict = OrderedDict((k, v) for k, v in checkpoint['state_dict'].items() if 'linear' not in k and 'fc' not in k)
        model.load_state_dict(new_state_dict, strict=False)
        print(f'===> Pretrained weights found in total: [{len(new_state_dict)}]')
        print(f'===> Pre-trained model loaded: {args.pretrained}')

    if args.resume:
        if os.path.isfile(args.resume):
            print(f"===> Loading checkpoint '{args.resume}'")
            checkpoint = torch.load(args.resume, map_location=device)
            args.start_epoch = checkpoint['epoch']
            args.best_loss = checkpoint['best_loss']
            model.load_state_dict(checkpoint['state_dict'])
            optimizer.load_state_dict(checkpoint['optimizer'])
            print(f"===> Loaded checkpoint '{args.resume}' (Epoch [{checkpoint['epoch']}])")
        else:
            print(f"===> No checkpoint found at '{args.resume}'")

    store = EpochFeatures(steps_per_epoch * args.batch_size, 2048, device) if args.fds else None
    for epoch in range(args.start_epoch, args.epoch):
        adjust_learning_rate(optimizer, epoch, args)
        if args.amp_switch_epoch is not None and args.amp == 'bf16':
            # the precision schedule: one engine, one graph; only the conv stack's arithmetic (and the dtype the GPU augmentation emits) changes
            want = None if epoch < args.amp_switch_epoch else torch.bfloat16
            if model.amp_dtype != want or epoch == args.start_epoch:
                model.set_amp_dtype(want)
                for aug in (locals().get('aug_train'), locals().get('aug_eval')):
                    if aug is not None:
                        aug.dtype = torch.float32 if want is None else torch.bfloat16
                print(f"Epoch [{epoch}]: conv stack in {('float32' if model.f32_arith == 'exact' else 'float32 on split-bf16 ' + model.f32_arith) if want is None else 'bf16'} (--amp_switch_epoch {args.amp_switch_epoch})")
        train_loss = train(train_batches(epoch), steps_per_epoch, model, optimizer, epoch, args, store)
        val_loss_mse, val_loss_l1, val_loss_gmean = validate(eval_batches(val_set), n_val, model, args, train_labels=train_labels)

        loss_metric = val_loss_mse if args.loss == 'mse' else val_loss_l1
        is_best = loss_metric < args.best_loss
        args.best_loss = min(loss_metric, args.best_loss)
        print(f"Best {'L1' if 'l1' in args.loss else 'MSE'} Loss: {args.best_loss:.3f}")
        if rank == 0:
            save_checkpoint(args, {'epoch': epoch + 1, 'model': args.model, 'best_loss': args.best_loss,
                                   'state_dict': model.state_dict(), 'optimizer': optimizer.state_dict()}, is_best)
        print(f"Epoch #{epoch}: Train loss [{train_loss:.4f}]; "
              f"Val loss: MSE [{val_loss_mse:.4f}], L1 [{val_loss_l1:.4f}], G-Mean [{val_loss_gmean:.4f}]")
        tb_logger.log_value('train_loss', train_loss, epoch)
        tb_logger.log_value('val_loss_mse', val_loss_mse, epoch)
        tb_logger.log_value('val_loss_l1', val_loss_l1, epoch)
        tb_logger.log_value('val_loss_gmean', val_loss_gmean, epoch)

    print("=" * 120)
    print("Test best model on testset...")
    if world > 1:
        torch.distributed.barrier()
    checkpoint = torch.load(f"{args.store_root}/{args.store_name}/ckpt.best.pth.tar", map_location=device)
    model.load_state_dict(checkpoint['state_dict'])
    print(f"Loaded best model, epoch {checkpoint['epoch']}, best val loss {checkpoint['best_loss']:.4f}")
    n_test = (len(test_set) + args.batch_size - 1) // args.batch_size
    test_loss_mse, test_loss_l1, test_loss_gmean = validate(eval_batches(test_set), n_test, model, args,
                                                            train_labels=train_labels, prefix='Test')
    print(f"Test loss: MSE [{test_loss_mse:.4f}], L1 [{test_loss_l1:.4f}], G-Mean [{test_loss_gmean:.4f}]\nDone")
