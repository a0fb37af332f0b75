"""Command-line flags of the block-copy engine (same names/defaults as the reference, core/argparser.py:1-12;
`fixed` is this repo's seeded fixed-fraction policy for reproducible benchmarks)."""

POLICIES = ["static", "all", "none", "random", "fixed", "rl_semseg", "rl_objectdetection"]

_FLAGS = [
    ("--block-policy", dict(type=str, default="rl_semseg", choices=POLICIES, help="policy name")),
    ("--block-num-classes", dict(type=int, default=19, help="number of output classes of the main task")),
    ("--block-optim-lr", dict(type=float, default=0.0001, help="policy learning rate")),
    ("--block-optim-wd", dict(type=float, default=0.001, help="policy weight decay")),
    ("--block-optim-momentum", dict(type=float, default=0, help="policy optimizer momentum")),
    ("--block-target", dict(type=float, default=0.50, help="target execution percentage")),
    ("--block-complexity-weight", dict(type=float, default=5, help="weight gamma, setting importance of complexity reward")),
    ("--block-size", dict(type=int, default=128, help="size of blocks in px")),
    ("--block-train-interval", dict(type=int, default=4, help="optimize the policy every N frames")),
    ("--block-cost-momentum", dict(type=float, default=0.9, help="cost momentum")),
    ("--block-policy-verbose", dict(action="store_true", help="print debug info for policy training")),
    ("--block-graph", dict(type=int, default=0, help="1: replay the packed pipeline as a hipGraph per executed-tile count")),
    ("--block-seed", dict(type=int, default=0, help="seed of the `fixed` policy's tile choice")),
]


def add_argparser_arguments(parser):
    for flag, kw in _FLAGS:
        parser.add_argument(flag, **kw)
    return parser


def default_settings(**overrides) -> dict:
    """The settings dict ``vars(args)`` would produce with all defaults."""
    import argparse

    ns = add_argparser_arguments(argparse.ArgumentParser()).parse_args([])
    d = vars(ns)
    d.update(overrides)
    return d
