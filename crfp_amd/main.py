"""Entry point with the reference's ``main.py`` dispatch (main.py:19-68) for the inference path:

    python -m crfp_amd.main --eval True --model_path <dir of checkpoints> --dataset_dir <REDS root> ...   (eval.sh's flags)

Same steps as the reference: experiment directory (utils.mkExpDir, utils.py:41-64), device from ``--cpu/--num_gpu/--gpu_id``
(main.py:27-30), ``CRFP.CRFP_DSV(mid_channels=32, y_only, hr_dcn, offset_prop, spynet_pretrained, device)`` (main.py:34), then
for every checkpoint of ``sorted(os.listdir(model_path))``: ``Trainer.load`` + ``Trainer.eval_basicvsr`` (main.py:56-62),
logging the reference's ``Ref  PSNR (now)`` lines.  Differences, on purpose: ``--cpu True`` is refused (this build has no
CPU path), ``--num_gpu N > 1`` means N ranks under ``torchrun`` sharding the clips (nn.DataParallel in the reference), and
training / ``--test`` media export are outside the hot path (OUT OF SCOPE).
"""
import logging
import os
import shutil
import sys

import torch


def mk_exp_dir(args):
    """utils.mkExpDir (utils.py:41-64): create save_dir (refuse / reset an existing one), dump args.txt, return a logger."""
    if os.path.exists(args.save_dir):
        if not args.reset:
            raise SystemExit('Error: save_dir "' + args.save_dir + '" already exists! Please set --reset True to delete the folder.')
        shutil.rmtree(args.save_dir)
    os.makedirs(args.save_dir)
    if (args.eval and args.eval_save_results) or args.test:
        os.makedirs(os.path.join(args.save_dir, 'save_results'))
    with open(os.path.join(args.save_dir, 'args.txt'), 'w') as f:
        for k, v in vars(args).items():
            f.write(k.rjust(30, ' ') + '\t' + str(v) + '\n')
    logger = logging.getLogger(args.logger_name)
    logger.setLevel(logging.DEBUG)
    fmt = logging.Formatter('[%(asctime)s] - [%(filename)s file line:%(lineno)d] - %(levelname)s: %(message)s')
    for h in (logging.FileHandler(os.path.join(args.save_dir, args.log_file_name)), logging.StreamHandler()):
        h.setFormatter(fmt)
        logger.addHandler(h)
    return logger


def select_device(args, local_rank=0):
    """main.py:27-30."""
    if args.cpu:
        raise SystemExit("crfp_amd: --cpu True is not supported: the product has no CPU path (the CPU restatement in "
                         "oracle/ is test infrastructure)")
    if args.num_gpu == 1:
        return torch.device('cuda:{}'.format(args.gpu_id))
    return torch.device('cuda', local_rank)


def build_model(args, device, spynet_pretrained='pretrained_models/fnet.pth'):
    """main.py:34.  A missing fnet.pth is tolerated (the checkpoint of --model_path carries the flow net as well)."""
    from .model import CRFP
    if spynet_pretrained is not None and not os.path.exists(spynet_pretrained):
        spynet_pretrained = None
    # The reference switches wirings by (un)commenting a factory line (main.py:34-35: CRFP_DSV is live, CRFP_DSV_CRA commented out);
    # here the class name comes from the environment, default the live line
    name = os.environ.get("CRFP_MODEL", "CRFP_DSV")
    if name not in ("CRFP_DSV", "CRFP_DSV_CRA", "CRFP", "CRFP_simple"):
        raise SystemExit(f"crfp_amd: CRFP_MODEL={name!r}: expected CRFP_DSV, CRFP_DSV_CRA, CRFP or CRFP_simple (classes of model/CRFP.py)")
    return getattr(CRFP, name)(mid_channels=32, y_only=args.y_only, hr_dcn=args.hr_dcn, offset_prop=args.offset_prop,
                               spynet_pretrained=spynet_pretrained, device=device).to(device)


def main(argv=None):
    from . import evalrig, option
    args = option.parse(argv)
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    if args.num_gpu > 1 and world != args.num_gpu:
        raise SystemExit(f"crfp_amd: --num_gpu {args.num_gpu} runs one process per GPU: launch with "
                         f"`torchrun --nproc-per-node {args.num_gpu} -m crfp_amd.main ...` (WORLD_SIZE is {world})")
    if not args.eval:
        raise SystemExit("crfp_amd.main runs the eval path (--eval True); training and --test media export are out of scope")
    if args.dataset.lower() != 'reds':
        raise SystemExit('Error: no such type of dataset!')
    logger = mk_exp_dir(args) if rank == 0 else logging.getLogger(args.logger_name)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")
    device = select_device(args, local)
    torch.cuda.set_device(device)
    model = build_model(args, device).eval()
    results = []
    for idx, m in enumerate(sorted(os.listdir(args.model_path))):   # main.py:57-61
        path = os.path.join(args.model_path, m)
        logger.info('load_model_path: ' + path)
        evalrig.load_checkpoint(model, path)
        res = evalrig.eval_reds(model, args, rank, world, dist, device)
        results.append(res)
        if rank == 0:   # trainer.py:383,397
            logger.info('Ref  PSNR (now): %.3f \t SSIM (now): %.4f' % (res["psnr"], res["ssim"]))
            logger.info('Ref  PSNR_Y (now): %.3f \t SSIM_Y (now): %.4f' % (res["psnr_y"], res["ssim_y"]))
    return results


if __name__ == '__main__':
    main(sys.argv[1:])
