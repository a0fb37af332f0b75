"""Settings of the block-copy engine.

The keys are the interface: model code builds them with ``add_argparser_arguments(parser)`` + ``vars(args)`` exactly as
with the reference (its core/argparser.py:1-12 defines the same flag names and defaults), or with ``default_settings()``
when there is no command line.  Two keys are additions of this implementation: ``block_graph`` and ``block_seed``.
"""
import argparse

POLICIES = ["static", "all", "none", "random", "fixed", "rl_semseg", "rl_objectdetection"]

# key (dashes become underscores) -> (python type or None for a boolean switch, default, what it controls)
_SPEC = {
    "block-policy": (str, "rl_semseg", "which tiles to execute: " + " | ".join(POLICIES)),
    "block-num-classes": (int, 19, "classes of the wrapped task (input width of the RL policy net)"),
    "block-optim-lr": (float, 1e-4, "RL policy: learning rate"),
    "block-optim-wd": (float, 1e-3, "RL policy: weight decay"),
    "block-optim-momentum": (float, 0, "RL policy: optimiser momentum"),
    "block-target": (float, 0.5, "fraction of tiles the policy should execute"),
    "block-complexity-weight": (float, 5, "RL policy: gamma of the execution-cost reward"),
    "block-size": (int, 128, "tile edge at input resolution, pixels"),
    "block-train-interval": (int, 4, "RL policy: one update every this many frames"),
    "block-cost-momentum": (float, 0.9, "RL policy: momentum of the running execution rate"),
    "block-policy-verbose": (None, False, "RL policy: print training diagnostics"),
    "block-graph": (int, 0, "1 = replay the packed pipeline as one hipGraph per executed-tile count"),
    "block-seed": (int, 0, "policy `fixed`: seed of the per-frame tile choice"),
}


def add_argparser_arguments(parser: argparse.ArgumentParser) -> argparse.ArgumentParser:
    for key, (typ, default, doc) in _SPEC.items():
        if typ is None:
            parser.add_argument("--" + key, action="store_true", help=doc)
        elif key == "block-policy":
            parser.add_argument("--" + key, type=typ, default=default, choices=POLICIES, help=doc)
        else:
            parser.add_argument("--" + key, type=typ, default=default, help=doc)
    return parser


def default_settings(**overrides) -> dict:
    """What ``vars(args)`` holds when no flag is given, with ``overrides`` applied on top."""
    settings = {key.replace("-", "_"): default for key, (_, default, _) in _SPEC.items()}
    settings.update(overrides)   # extra keys pass through (model code may carry its own entries in the same dict)
    return settings
