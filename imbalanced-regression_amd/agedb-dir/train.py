"""Drop-in for the reference's ``agedb-dir/train.py`` — same command line (defaults of this sub-project:
dataset=agedb, bucket_start=3), one process per MI355X:

    python train.py --reweight sqrt_inv --lds --lds_kernel gaussian --lds_ks 5 --lds_sigma 2 --fds ...
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...   # 8 GPUs
    python train.py --synthetic 4096 --fds --lds --reweight sqrt_inv --epoch 3                           # no image files
"""
import _path  # noqa: F401
from dirhip.train_main import run, shot_metrics, train, validate  # noqa: F401
from loss import *  # noqa: F401,F403
from utils import *  # noqa: F401,F403

if __name__ == '__main__':
    run(dataset_default='agedb')
