#!/usr/bin/env python
"""Training entry point with the reference's command line (reference train.py:1-70):

    python train.py -c examples/csmsc/configs/msmc_vq_gan.yaml                      # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py -c <yaml> -g dp

One process per GPU.  Under ``torch.distributed.run`` rank and world size come from the environment (RCCL over xGMI);
the reference's own launcher arguments (``-r rank -g group``, one process started per rank by hand with the
``distributed.dist_url`` of the YAML) work too.  Like the reference, the global ``dataloader.batch_size`` is divided by the
number of ranks.  ``--graphs`` replays the GAN-phase step as hipGraphs (needs a fixed ``dataset.segment_length``).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, 'msmc-tts_amd')]
import msmctts_amd  # noqa: E402,F401  (before the first HIP call: runtime switches, see its docstring)
import torch  # noqa: E402

from msmctts_amd.distributed.distributed import init_distributed  # noqa: E402
from msmctts_amd.tasks import build_task  # noqa: E402
from msmctts_amd.trainers import build_trainer  # noqa: E402
from msmctts_amd.utils.config import Config  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('-c', '--config', required=True, help='YAML file for configuration')
    ap.add_argument('-r', '--rank', type=int, default=int(os.environ.get('RANK', '0')), help='rank of this process')
    ap.add_argument('-g', '--group_name', default='', help='name of the process group (any non-empty string: distributed)')
    ap.add_argument('--graphs', action='store_true', help='replay the GAN-phase step as hipGraphs')
    ap.add_argument('--bf16', action='store_true', help='bf16 convolution / GEMM bodies (fp32 VQ search and spectra)')
    args = ap.parse_args()
    config = Config(args.config)
    if not config.save_checkpoint_dir:
        config.save_checkpoint_dir = os.path.join(os.path.dirname(args.config), 'checkpoints')
    world = int(os.environ.get('WORLD_SIZE', '0')) or torch.cuda.device_count()
    if world > 1 and args.group_name == '' and 'WORLD_SIZE' not in os.environ:
        print('WARNING: Multiple GPUs detected but no distributed group set')
        world = 1
    if world == 1 and args.rank != 0:
        raise SystemExit('Doing single GPU training on rank > 0')
    torch.manual_seed(config.seed)
    if world > 1:
        url = 'env://' if 'MASTER_ADDR' in os.environ else config.distributed.dist_url
        init_distributed(args.rank, world, args.group_name, config.distributed.dist_backend, url)
        config.dataloader.batch_size = config.dataloader.batch_size // world
        print('Batch size per GPU is changed to {}.'.format(config.dataloader.batch_size))
    task = build_task(config, 'train')
    trainer = build_trainer(config, task, num_gpus=world, rank=args.rank)
    trainer.use_graphs = args.graphs
    if args.bf16:
        trainer.amp_dtype = torch.bfloat16
    trainer.train()
    print('Training done!')


if __name__ == '__main__':
    main()
