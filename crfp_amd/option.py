"""Command-line surface of the reference's ``option.py`` (option.py:10-119) for the inference path.

Every flag of the reference is accepted with the same name, type and default, so ``eval.sh``'s argument list parses
unchanged.  Flags that only steer training, Visdom or the model variants this build does not ship are parsed and then
ignored (listed in ``IGNORED``); the ones the eval path reads are in ``USED``.  Unlike the reference, nothing is parsed
at import time: call ``parse(argv)``.
"""
import argparse


def str2bool(v):
    """option.py:3-9."""
    if isinstance(v, bool):
        return v
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Boolean value expected.')


# (flag, type, default) in the reference's order
_FLAGS = [
    ("visdom_port", int, 8801), ("visdom_view", str, "MRCF"),
    ("save_dir", str, "save_dir"), ("reset", str2bool, False), ("log_file_name", str, "MRCF.log"),
    ("logger_name", str, "MRCF"),
    ("cpu", str2bool, False), ("num_gpu", int, 1), ("gpu_id", int, 0),
    ("dataset", str, "REDS"), ("dataset_dir", str, "/Data/REDS_sharp/"),
    ("num_workers", int, 4),
    ("num_res_blocks", str, "4+4+4+4"), ("n_feats", int, 64), ("res_scale", float, 1.0), ("cra", str2bool, True),
    ("mrcf", str2bool, True), ("y_only", str2bool, False), ("hr_dcn", str2bool, True), ("offset_prop", str2bool, True),
    ("rec_w", float, 1.0),
    ("beta1", float, 0.9), ("beta2", float, 0.999), ("eps", float, 1e-12), ("lr_rate", float, 1e-4),
    ("lr_rate_flow", float, 2.5e-5), ("decay", float, 999999), ("gamma", float, 0.5),
    ("batch_size", int, 8), ("GT_size", int, 256), ("FV_size", int, 80), ("scale", int, 4), ("N_frames", int, 15),
    ("train_crop_size", int, 40), ("num_init_epochs", int, 2), ("num_epochs", int, 1), ("print_every", int, 1),
    ("save_every", int, 999999), ("val_every", int, 999999),
    ("eval", str2bool, False), ("eval_save_results", str2bool, False), ("model_path", str, None),
    ("test", str2bool, False),
]

USED = ("save_dir", "reset", "log_file_name", "logger_name", "cpu", "num_gpu", "gpu_id", "dataset", "dataset_dir",
        "y_only", "hr_dcn", "offset_prop", "GT_size", "FV_size", "scale", "N_frames", "eval", "model_path", "test")
IGNORED = tuple(f for f, _, _ in _FLAGS if f not in USED)


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="MRCF")
    for name, typ, default in _FLAGS:
        p.add_argument("--" + name, type=typ, default=default)
    return p


def parse(argv=None):
    return build_parser().parse_args(argv)
