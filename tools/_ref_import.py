"""Import shim for the upstream reference (runs ONLY in the build container).

The reference at /root/reference is pure Python on top of PyTorch.  Six third-party
modules it imports at module scope are absent from this image and perform no
hot-path arithmetic; they are replaced by inert stand-ins so that the
reference's model/handler code can be imported and executed on CPU to generate
golden vectors.  Nothing in here travels to the GPU box as a dependency: tests read
only the .npz/.json fixtures this produces.
"""
import collections
import collections.abc
import sys
import types

REF_CODE = "/root/reference/Code"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    if getattr(install, "_done", False):
        return
    # python >= 3.10 removed the collections.Callable alias used by helper_functions.py:5
    collections.Callable = collections.abc.Callable

    import tomli

    def _toml_load(f):
        if isinstance(f, (str, bytes)):
            with open(f, "rb") as fh:
                return tomli.load(fh)
        data = f.read()
        if isinstance(data, str):
            data = data.encode()
        return tomli.loads(data.decode())

    _mod("toml", load=_toml_load, loads=tomli.loads, dump=lambda *a, **k: None)

    class _Fore:
        def __getattr__(self, k):
            return ""

    _mod("colorama", init=lambda *a, **k: None, Fore=_Fore())
    _mod("moviepy")
    _mod("moviepy.video")
    _mod("moviepy.video.io")
    _mod("moviepy.video.io.ImageSequenceClip", ImageSequenceClip=None)

    import numpy as np
    import torch

    class ToTensor:
        def __call__(self, pic):
            arr = np.asarray(pic)
            if arr.ndim == 2:
                arr = arr[:, :, None]
            t = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
            if t.dtype == torch.uint8:
                return t.to(torch.float32).div(255)
            return t

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class _Unused:
        def __init__(self, *a, **k):
            pass

    tv = _mod("torchvision")
    tv.transforms = _mod("torchvision.transforms", ToTensor=ToTensor, Compose=Compose,
                         Normalize=_Unused, ToPILImage=_Unused)
    tv.models = _mod("torchvision.models", vgg19=None)
    sk = _mod("skimage")
    sk.metrics = _mod("skimage.metrics", structural_similarity=lambda *a, **k: float("nan"))
    sk.io = _mod("skimage.io", imsave=lambda *a, **k: None)
    _mod("click_config_file", configuration_option=lambda *a, **k: (lambda f: f))
    if REF_CODE not in sys.path:
        sys.path.insert(0, REF_CODE)
    install._done = True
