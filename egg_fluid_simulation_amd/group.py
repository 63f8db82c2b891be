"""SimulationGroup -- several GPUs of ONE process behind the SimulationHandler surface (ctypes twin of the egg_group_*
entry points of include/eggsim.h; csrc/eggsim_group.cpp).  One device handle per x-slab, global batch ids, batches
handed between devices when their claims meet across a cut; results equal a single handler's bit for bit.
(Between processes -- one per GPU, RCCL -- the same protocol is sharding.ShardedSimulationHandler.)"""
import ctypes as C
import math

import numpy as np

from . import _ffi
from .simulation_handler import EggError, SimulationHandler


class _Borrowed(SimulationHandler):
    """a device handle owned by the group: downloads and statistics only, never destroyed from here"""

    def __init__(self, lib, ptr):  # (no egg_create)
        self._lib, self._h = lib, C.c_void_p(ptr)

    def close(self):
        self._h = None


class SimulationGroup:
    def __init__(self, devices, cuts=None, white_config=None, yolk_config=None):
        tmpl = SimulationHandler.__new__(SimulationHandler)  # config validation of the reference (L:1253-1320), no device
        from .default_config import default_configs
        import copy
        if white_config is None and yolk_config is None:
            white_config, yolk_config = default_configs()
        if yolk_config is None:
            yolk_config = white_config
        tmpl._white_config, tmpl._yolk_config = {}, {}
        tmpl._mass_distribution_variance, tmpl._max_collision_fraction = 4, 0.05
        tmpl._load_config(copy.deepcopy(white_config), True)
        tmpl._load_config(copy.deepcopy(yolk_config), False)
        self._lib = _ffi.load()
        devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        cut = None
        if cuts is not None:
            if len(cuts) != len(devices) + 1:
                raise EggError("[ERROR] In SimulationGroup.new: need len(devices) + 1 cuts")
            cut = (C.c_double * len(cuts))(*[float(c) for c in cuts])
        g = C.c_void_p()
        rc = self._lib.egg_group_create(C.byref(tmpl._c_config(True)), C.byref(tmpl._c_config(False)), len(devices), devs, cut, C.byref(g))
        if rc != _ffi.EGG_OK:
            raise EggError("[ERROR] In SimulationGroup.new: " + self._lib.egg_last_error(None).decode())
        self._g = g
        self.handles = [_Borrowed(self._lib, self._lib.egg_group_handle(self._g, k)) for k in range(len(devices))]

    def __del__(self):
        self.close()

    def close(self):
        if getattr(self, "_g", None):
            for h in self.handles:
                h.close()
            self._lib.egg_group_destroy(self._g)
            self._g = None

    def _check(self, rc):
        if rc < 0:
            raise EggError("[ERROR] " + self._lib.egg_group_last_error(self._g).decode())
        return rc

    def add(self, x, y, white_radius=None, yolk_radius=None, white_n=None, yolk_n=None):
        out = C.c_int64()
        self._check(self._lib.egg_group_add(
            self._g, float(x), float(y), float("nan") if white_radius is None else float(white_radius),
            float("nan") if yolk_radius is None else float(yolk_radius),
            _ffi.DEFAULT_COUNT if white_n is None else int(white_n), _ffi.DEFAULT_COUNT if yolk_n is None else int(yolk_n), C.byref(out)))
        return out.value

    def remove(self, batch_id):
        self._check(self._lib.egg_group_remove(self._g, int(batch_id)))

    def set_target_position(self, batch_id, x, y):
        self._check(self._lib.egg_group_set_target(self._g, int(batch_id), float(x), float(y)))

    def get_position(self, batch_id):
        x, y = C.c_double(), C.c_double()
        self._check(self._lib.egg_group_get_position(self._g, int(batch_id), C.byref(x), C.byref(y)))
        return x.value, y.value

    def update(self, delta, step_delta=None, n_substeps=None, n_collision_steps=None):
        n = C.c_int32()
        self._check(self._lib.egg_group_update(self._g, float(delta), 1 / 60 if step_delta is None else float(step_delta),
                                               2 if n_substeps is None else int(math.ceil(n_substeps)),
                                               3 if n_collision_steps is None else int(math.ceil(n_collision_steps)), C.byref(n)))
        return n.value

    def step(self, delta=1 / 60, n_substeps=2, n_collision_steps=3):
        self._check(self._lib.egg_group_step(self._g, float(delta), int(n_substeps), int(n_collision_steps)))

    def owner(self, batch_id):
        k, lid = C.c_int32(), C.c_int64()
        if self._lib.egg_group_owner(self._g, int(batch_id), C.byref(k), C.byref(lid)) != _ffi.EGG_OK:
            raise EggError("[ERROR] In SimulationGroup.owner: no batch with id `%s`" % batch_id)
        return k.value, lid.value

    def counters(self):
        m, d = C.c_int64(), C.c_int64()
        self._check(self._lib.egg_group_get_counters(self._g, C.byref(m), C.byref(d)))
        return dict(migrations=m.value, discarded_steps=d.value)

    def particles(self, which):
        """{global id: (x[n], y[n])} over all devices (the batches of a handle are laid out in ascending global id)"""
        out = {}
        owners = {}
        gid = 1
        while True:
            try:
                owners[gid] = self.owner(gid)
            except EggError:
                break
            gid += 1
        for k, h in enumerate(self.handles):
            mine = sorted(g for g, (dev, _l) in owners.items() if dev == k)
            if not mine:
                continue
            x, y = h.download(which, "x"), h.download(which, "y")
            off = 0
            for g in mine:
                nw, ny = h.get_n_particles(owners[g][1])
                n = nw if which == _ffi.WHITE else ny
                out[g] = (x[off:off + n], y[off:off + n])
                off += n
        return out
