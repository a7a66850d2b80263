"""Base class of the conv-stack modules: holds parameters under the reference's key names and
keeps the packed HIP weight blob in sync with them."""
import torch.nn as nn

from ..netspec import ParamTree


class PackedNet(ParamTree):
    """A ParamTree whose parameters feed one packed blob of the shared ``HotPath``.

    The blob is re-packed (``fvp_pack_conv`` per conv, on the GPU) the first time the
    module runs after construction, ``load_state_dict``, or ``.to()/.cuda()/.float()``.
    Call ``mark_dirty()`` after modifying parameters in place by hand."""

    def _init_packing(self, engine, stack_name):
        self.__dict__["engine"] = engine          # plain attribute: not a submodule, not in state_dict
        self.__dict__["stack_name"] = stack_name
        self.__dict__["_dirty"] = True
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.mark_dirty())

    def mark_dirty(self):
        self.__dict__["_dirty"] = True

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self.mark_dirty()
        return out

    def _pack(self):
        self.engine.pack_stack(self.stack_name, self)

    def ensure_packed(self):
        if self._dirty:
            self._pack()
            self.__dict__["_dirty"] = False
