"""blockcopy -- MI355X-native block-sparse video inference behind the reference package's Python API.

``import blockcopy`` resolves to this package when ``blockcopy-video-processing-pytorch_amd/`` precedes the reference
on ``sys.path``; model code written against thomasverelst/blockcopy-video-processing-pytorch keeps working
(its blockcopy/blockcopy/__init__.py:1-4 exports the same names):

    model = blockcopy.BlockCopyModel(net, settings=vars(args))     # wrap
    blockcopy.add_argparser_arguments(parser)                      # --block-* flags
    @blockcopy.blockcopy_noblocks                                  # run a submodule on dense maps
    blockcopy.to_tensorwrapper(x) / to_tensor(x) / is_block(x) / is_tensorwrapper(x)

All device work goes through ``lib/libblockcopy_hip.so`` (C ABI: ``include/blockcopy_hip.h``); there is no CPU fallback.
"""
from blockcopy.core.argparser import add_argparser_arguments
from blockcopy.core.blockcopy import BlockCopyModel, blockcopy_noblocks
from blockcopy.core.tensorwrapper import (TensorWrapper, is_block, is_tensorwrapper, to_tensor,
                                          to_tensorwrapper)
from blockcopy.policy.policy import build_policy_from_settings

__all__ = [
    "BlockCopyModel", "blockcopy_noblocks", "add_argparser_arguments", "build_policy_from_settings",
    "TensorWrapper", "to_tensorwrapper", "to_tensor", "is_block", "is_tensorwrapper",
]
