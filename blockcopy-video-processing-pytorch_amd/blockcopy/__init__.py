"""blockcopy -- MI355X-native block-sparse video-inference engine with the reference package's Python API.

Drop-in for the import name ``blockcopy`` of thomasverelst/blockcopy-video-processing-pytorch
(blockcopy/blockcopy/__init__.py:1-4)."""
from blockcopy.core.tensorwrapper import TensorWrapper, is_block, is_tensorwrapper, to_tensorwrapper, to_tensor
from blockcopy.core.blockcopy import BlockCopyModel, blockcopy_noblocks
from blockcopy.core.argparser import add_argparser_arguments
from blockcopy.policy.policy import build_policy_from_settings

__all__ = ["TensorWrapper", "is_block", "is_tensorwrapper", "to_tensorwrapper", "to_tensor", "BlockCopyModel",
           "blockcopy_noblocks", "add_argparser_arguments", "build_policy_from_settings"]
