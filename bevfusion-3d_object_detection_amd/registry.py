"""Plug-in registration under the reference's registry names.

The reference finds its modules by NAME through mmengine's registry
(`@MODELS.register_module()`, mmdet3d/registry.py:71-72; names used by the BEVFusion configs:
'DepthLSSTransform', 'LSSTransform', 'BEVFusionSparseEncoder', 'SubMConv3d', 'SparseConv3d', ...).
mmengine/mmdet3d are not installed in this image, so a minimal registry with the same two calls
(`register_module`, `build`) lives here; `register(target)` re-registers every class into a real
mmengine registry (e.g. mmdet3d.registry.MODELS) when one is available -- that one call is the
whole integration on the reference side.
"""


class Registry:
    def __init__(self, name):
        self.name = name
        self._modules = {}

    def register_module(self, name=None, force=False, module=None):
        def _do(cls):
            key = name or cls.__name__
            if key in self._modules and not force and self._modules[key] is not cls:
                raise KeyError("%s is already registered in %s" % (key, self.name))
            self._modules[key] = cls
            return cls
        if module is not None:
            return _do(module)
        return _do

    def get(self, key):
        return self._modules.get(key)

    def build(self, cfg, *args, **kwargs):
        cfg = dict(cfg)
        typ = cfg.pop("type")
        cls = self._modules[typ] if isinstance(typ, str) else typ
        return cls(*args, **cfg, **kwargs)

    def __contains__(self, key):
        return key in self._modules

    def names(self):
        return sorted(self._modules)


MODELS = Registry("models")


def register(target):
    """Register every class of this package into `target` (an mmengine Registry) under the
    reference's names, overriding the CUDA-backed originals."""
    for key, cls in MODELS._modules.items():
        target.register_module(name=key, force=True, module=cls)
    return target
