"""MI355X-native SISR forward/backward engine (gfx950 HIP kernels behind the reference's model API).

Import with ``importlib.import_module("super-resolution-meta-attention-networks_amd")`` or via the
``sisr_amd`` shim at the repository root.  Sub-modules: hip (C-ABI binding), ops (autograd operators),
architectures (drop-in nn.Modules), handlers (model-handler API), parallel (RCCL data parallelism),
metrics (PSNR).
"""
from . import hip, ops, optim, architectures, metrics, handlers, parallel, han, san, srmd, sftmd, sparnet, degrade, data, cli  # noqa: F401
from .handlers import ModelInterface, available_models  # noqa: F401
