"""Host-side mirror of the reference's `SimulationHandler` class.

Same method names, argument meaning, defaults, warnings and errors as
/root/reference/simulation_handler.lua:9-459 ("L:" below), with the particle
state and the whole `_step` on the MI355X behind libeggsim.so (include/eggsim.h).
The LuaJIT twin of this file is lua/egg_fluid_simulation/simulation_handler.lua.

Error conventions (log.lua:9-88): where the reference calls `log.error` /
`log.assert` this raises `EggError`; where it calls `log.warning` this issues an
`EggWarning` through the `warnings` module and carries on.
"""
import copy
import ctypes as C
import math
import warnings

import numpy as np

from . import _ffi
from .default_config import default_configs


class EggError(RuntimeError):
    """The reference's `log.error` (a thrown Lua error)."""


class EggWarning(UserWarning):
    """The reference's `log.warning` (a line on stderr)."""


def _type_name(v):
    if isinstance(v, bool):
        return "boolean"
    if isinstance(v, (int, float, np.integer, np.floating)):
        return "number"
    if v is None:
        return "nil"
    if isinstance(v, (dict, list, tuple)):
        return "table"
    if isinstance(v, str):
        return "string"
    return type(v).__name__


def _assert_types(*pairs):  # log.assert, log.lua:65-88
    for k in range(0, len(pairs), 2):
        value, expected = pairs[k], pairs[k + 1]
        if _type_name(value) != expected:
            raise EggError("[ERROR] for argument #%d: expected `%s`, got `%s`"
                           % (k // 2 + 1, expected, _type_name(value)))


def _is_nan(x):
    return x != x


# L:1152-1249: key -> (type, min, max)
_VALID_CONFIG_KEYS = {
    "damping": ("number", 0, 1),
    "color": ("color", None, None),
    "outline_color": ("color", None, None),
    "outline_thickness": ("number", 0, None),
    "collision_strength": ("number", 0, 1),
    "collision_overlap_factor": ("number", 0, None),
    "cohesion_strength": ("number", 0, 1),
    "cohesion_interaction_distance_factor": ("number", 0, None),
    "follow_strength": ("number", 0, 1),
    "min_radius": ("number", 0, None),
    "max_radius": ("number", 0, None),
    "min_mass": ("number", 0, None),
    "max_mass": ("number", 0, None),
    "motion_blur": ("number", 0, 1),
    "texture_scale": ("number", 1, None),
    "highlight_strength": ("number", 0, None),
    "shadow_strength": ("number", 0, None),
}

_SOLVER_KEYS = ["damping", "follow_strength", "cohesion_strength", "cohesion_interaction_distance_factor",
                "collision_strength", "collision_overlap_factor", "min_mass", "max_mass", "min_radius",
                "max_radius"]


class SimulationHandler:
    """`SimulationHandler(white_config, yolk_config)` (L:11-15, L:425-459)."""

    def __init__(self, white_config=None, yolk_config=None, device=0):
        if white_config is None and yolk_config is None:
            white_config, yolk_config = default_configs()
        if yolk_config is None:  # L:426
            yolk_config = white_config
        _assert_types(white_config, "table", yolk_config, "table")
        self._lib = _ffi.load()
        self._h = None
        self._white_config = {}
        self._yolk_config = {}
        self._load_config(copy.deepcopy(white_config), True)
        self._load_config(copy.deepcopy(yolk_config), False)
        # hidden constants (L:447-448, math.lua:2)
        self._mass_distribution_variance = 4
        self._max_collision_fraction = 0.05
        self._batch_colors = {}
        # render constants (L:444-449)
        self._thresholding_threshold = 0.3
        self._thresholding_smoothness = 0.01
        self._use_particle_color_flag = False
        self._use_lighting_flag = True
        h = C.c_void_p()
        rc = self._lib.egg_create(C.byref(self._c_config(True)), C.byref(self._c_config(False)), int(device),
                                  C.byref(h))
        if rc != _ffi.EGG_OK:
            raise EggError("[ERROR] In SimulationHandler.new: " + self._lib.egg_last_error(None).decode())
        self._h = h
        self._send_render_config()

    def __del__(self):
        self.close()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.egg_destroy(self._h)
            self._h = None

    # ------------------------------------------------------------------ config
    def _load_config(self, config, white_or_yolk):  # L:1253-1320
        scope = "In SimulationHandler.set_white_config: " if white_or_yolk else "In SimulationHandler.set_yolk_config: "
        target = self._white_config if white_or_yolk else self._yolk_config
        for key, value in config.items():
            entry = _VALID_CONFIG_KEYS.get(key)
            if entry is None:
                warnings.warn(scope + "unrecognized config key `%s`, it will be ignored" % key, EggWarning)
                continue
            typ, lo, hi = entry
            if typ == "color":
                value = list(value)
                if len(value) != 4:
                    raise EggError("[ERROR] " + scope + "color `%s` does not have 4 components" % key)
                for i in range(4):
                    component = value[i]
                    if _type_name(component) != "number" or _is_nan(component):
                        raise EggError("[ERROR] " + scope + "color `%s` has a component that is not a number" % key)
                    if component < 0 or component > 1:
                        warnings.warn(scope + "color `%s` has a component that is outside of [0, 1]" % key, EggWarning)
                    value[i] = min(max(component, 0), 1)
            else:
                if _type_name(value) != typ:
                    raise EggError("[ERROR] " + scope + "wrong type for config key `%s`, expected `%s`, got `%s`"
                                   % (key, typ, _type_name(value)))
                if _is_nan(value):
                    warnings.warn(scope + "config key `%s` is NaN, it will be ignored" % key, EggWarning)
                    continue
                if lo is not None and value < lo:
                    warnings.warn(scope + "config key `%s`'s value is `%s`, expected a value larger than `%s`"
                                  % (key, value, lo), EggWarning)
                    value = max(value, lo)
                elif hi is not None and value > hi:
                    warnings.warn(scope + "config key `%s`'s value is `%s`, expected a value smaller than `%s`"
                                  % (key, value, hi), EggWarning)
                    value = min(value, hi)
            target[key] = value

    def _c_config(self, white_or_yolk):
        cfg = self._white_config if white_or_yolk else self._yolk_config
        c = _ffi.EggConfig()
        for k in _SOLVER_KEYS:
            if k not in cfg:
                raise EggError("[ERROR] In SimulationHandler.new: config key `%s` is missing" % k)
            setattr(c, k, float(cfg[k]))
        c.max_collision_fraction = getattr(self, "_max_collision_fraction", 0.05)
        c.mass_distribution_variance = getattr(self, "_mass_distribution_variance", 4)
        c.eps = 1e-8
        return c

    _RENDER_DEFAULTS = dict(outline_thickness=1.0, highlight_strength=0.0, shadow_strength=0.0, texture_scale=12.0,
                            motion_blur=0.0003)

    def _send_render_config(self):
        # this object's config tables are the authority for the render keys (colour tables may be shared with batches,
        # L:49-50); the library gets a copy whenever they may have changed
        for which in range(2):
            self._check(self._lib.egg_set_render_config(self._h, which, C.byref(self._c_render_config(which == 0))))

    def _c_render_config(self, white_or_yolk):
        """the render keys of a config table (simulation_handler_default_config.lua:22-36) as egg_render_config"""
        cfg = self._white_config if white_or_yolk else self._yolk_config
        c = _ffi.EggRenderConfig()
        c.color[:] = [float(v) for v in cfg.get("color", [1, 1, 1, 1])]
        c.outline_color[:] = [float(v) for v in cfg.get("outline_color", [1, 1, 1, 1])]
        for k, d in self._RENDER_DEFAULTS.items():
            setattr(c, k, float(cfg.get(k, d)))
        return c

    # the handler's private render switches (L:448-449): particles take their batch's colour at `add` only while
    # _use_particle_color is set (L:978-990)
    @property
    def _use_particle_color(self):
        return self._use_particle_color_flag

    @_use_particle_color.setter
    def _use_particle_color(self, flag):
        self._use_particle_color_flag = bool(flag)
        self._check(self._lib.egg_set_render_flags(self._h, int(self._use_particle_color_flag), int(self._use_lighting_flag)))

    @property
    def _use_lighting(self):
        return self._use_lighting_flag

    @_use_lighting.setter
    def _use_lighting(self, flag):
        self._use_lighting_flag = bool(flag)
        self._check(self._lib.egg_set_render_flags(self._h, int(self._use_particle_color_flag), int(self._use_lighting_flag)))

    def set_white_config(self, config):  # L:226-229
        _assert_types(config, "table")
        self._load_config(copy.deepcopy(config), True)
        self._check(self._lib.egg_set_config(self._h, _ffi.WHITE, C.byref(self._c_config(True))))
        self._send_render_config()

    def set_yolk_config(self, config):  # L:233-236
        _assert_types(config, "table")
        self._load_config(copy.deepcopy(config), False)
        self._check(self._lib.egg_set_config(self._h, _ffi.YOLK, C.byref(self._c_config(False))))
        self._send_render_config()

    def get_white_config(self):  # L:240-242
        return copy.deepcopy(self._white_config)

    def get_yolk_config(self):  # L:246-248
        return copy.deepcopy(self._yolk_config)

    # ------------------------------------------------------------ error mapping
    def _message(self):
        return self._lib.egg_last_error(self._h).decode()

    def _check(self, rc):
        if rc == _ffi.EGG_OK:
            return rc
        if rc > 0:  # warning class: the reference prints and carries on
            warnings.warn(self._message(), EggWarning)
            return rc
        raise EggError("[ERROR] " + self._message())

    # --------------------------------------------------------------------- add
    def add(self, x, y, white_radius=None, yolk_radius=None, white_color=None, yolk_color=None,
            white_n_particles=None, yolk_n_particles=None):  # L:27-135
        _assert_types(x, "number", y, "number")
        given = (white_color is not None, yolk_color is not None)
        white_color = white_color if white_color is not None else self._white_config.get("color", [1, 1, 1, 1])
        yolk_color = yolk_color if yolk_color is not None else self._yolk_config.get("color", [1, 1, 1, 1])
        for v in (white_radius, yolk_radius, white_n_particles, yolk_n_particles):
            if v is not None:
                _assert_types(v, "number")
        _assert_types(white_color, "table", yolk_color, "table")
        # L:71-85 come before the colour checks in the reference; nothing may be created when they fail, so they
        # are made here before the C call (the library repeats them: only EGG_DEFAULT_COUNT means "not given")
        if white_radius is not None and white_radius <= 0:
            raise EggError("[ERROR] In SimulationHandler.add: white radius cannot be 0 or negative")
        if yolk_radius is not None and yolk_radius <= 0:
            raise EggError("[ERROR] In SimulationHandler.add: yolk radius cannot be 0 or negative")
        if white_n_particles is not None and white_n_particles <= 1:
            raise EggError("[ERROR] In SimulationHandler.add: white particle count cannot be 1 or negative")
        if yolk_n_particles is not None and yolk_n_particles <= 1:
            raise EggError("[ERROR] In SimulationHandler.add: yolk particle count cannot be 1 or negative")
        for name, color in (("white", white_color), ("yolk", yolk_color)):  # L:87-108
            for i, cname in enumerate("rgba"):
                if i >= len(color) or _type_name(color[i]) != "number" or _is_nan(color[i]):
                    raise EggError("[ERROR] In SimulationHandler.add: %s color component `%s` is not a number"
                                   % (name, cname))
                if color[i] < 0 or color[i] > 1:
                    warnings.warn("In SimulationHandler.add: %s color component `%s` is outside of [0, 1]"
                                  % (name, cname), EggWarning)
        out = C.c_int64()
        rc = self._lib.egg_add(self._h, float(x), float(y),
                               float("nan") if white_radius is None else float(white_radius),
                               float("nan") if yolk_radius is None else float(yolk_radius),
                               _ffi.DEFAULT_COUNT if white_n_particles is None else int(math.ceil(white_n_particles)),
                               _ffi.DEFAULT_COUNT if yolk_n_particles is None else int(math.ceil(yolk_n_particles)),
                               C.byref(out))
        self._check(rc)
        # L:49-50, L:124-129: a batch created without a colour shares the CONFIG's colour table (set_*_color on it then
        # changes config.color too); the device library is told which tables are the batch's own
        self._batch_colors[out.value] = [white_color, yolk_color]
        for which, color in enumerate((white_color, yolk_color)):
            if given[which]:
                self._lib.egg_set_add_color(self._h, out.value, which, *[float(c) for c in color[:4]])
        return out.value

    def add_many(self, xs, ys, white_radius=None, yolk_radius=None, white_n_particles=None,
                 yolk_n_particles=None):
        """Bulk form of `add` for 10^4..10^5 batches (one C call); returns the ids."""
        xs = np.ascontiguousarray(xs, dtype=np.float64)
        ys = np.ascontiguousarray(ys, dtype=np.float64)
        if xs.shape != ys.shape or xs.ndim != 1:
            raise EggError("[ERROR] In SimulationHandler.add_many: xs and ys must be 1-d arrays of equal length")
        ids = np.empty(xs.shape[0], dtype=np.int64)
        self._check(self._lib.egg_add_many(
            self._h, xs.shape[0], xs.ctypes.data, ys.ctypes.data,
            float("nan") if white_radius is None else float(white_radius),
            float("nan") if yolk_radius is None else float(yolk_radius),
            _ffi.DEFAULT_COUNT if white_n_particles is None else int(white_n_particles),
            _ffi.DEFAULT_COUNT if yolk_n_particles is None else int(yolk_n_particles), ids.ctypes.data))
        return ids

    def add_many_keyed(self, xs, ys, keys, white_radius=None, yolk_radius=None):
        """`add_many` with explicit global-order keys (multi-GPU sharding, see include/eggsim.h)"""
        xs = np.ascontiguousarray(xs, dtype=np.float64)
        ys = np.ascontiguousarray(ys, dtype=np.float64)
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        ids = np.empty(xs.shape[0], dtype=np.int64)
        self._check(self._lib.egg_add_many_keyed(
            self._h, xs.shape[0], xs.ctypes.data, ys.ctypes.data,
            float("nan") if white_radius is None else float(white_radius),
            float("nan") if yolk_radius is None else float(yolk_radius), _ffi.DEFAULT_COUNT, _ffi.DEFAULT_COUNT,
            keys.ctypes.data, ids.ctypes.data))
        return ids

    def export_batch(self, batch_id):
        """(info dict, white_state[9, n_w], yolk_state[9, n_y]) of a batch: everything another handler
        needs to continue it bit for bit"""
        nw, ny = self.get_n_particles(batch_id)
        info = _ffi.EggBatchInfo()
        ws, ys = np.empty((9, nw)), np.empty((9, ny))
        self._check(self._lib.egg_export_batch(self._h, int(batch_id), C.byref(info), ws.ctypes.data, ys.ctypes.data))
        return {k: getattr(info, k) for k, _ in _ffi.EggBatchInfo._fields_}, ws, ys

    def export_batch_to(self, batch_id, white_ptr, yolk_ptr):
        """export_batch into caller-owned buffers given by ADDRESS (host memory or memory of this handle's device, e.g.
        torch.Tensor.data_ptr() of a CUDA tensor: a device-to-device hand-over, nothing touches the host); returns info"""
        info = _ffi.EggBatchInfo()
        self._check(self._lib.egg_export_batch(self._h, int(batch_id), C.byref(info), C.c_void_p(int(white_ptr)), C.c_void_p(int(yolk_ptr))))
        return {k: getattr(info, k) for k, _ in _ffi.EggBatchInfo._fields_}

    def import_batch_from(self, info, white_ptr, yolk_ptr):
        """import_batch from buffers given by address (host or this handle's device memory, field-major [9, n])"""
        c = _ffi.EggBatchInfo(**{k: info[k] for k, _ in _ffi.EggBatchInfo._fields_})
        out = C.c_int64()
        self._check(self._lib.egg_import_batch(self._h, C.byref(c), C.c_void_p(int(white_ptr)), C.c_void_p(int(yolk_ptr)), C.byref(out)))
        return out.value

    def import_batch(self, info, white_state, yolk_state):
        c = _ffi.EggBatchInfo(**{k: info[k] for k, _ in _ffi.EggBatchInfo._fields_})
        ws = np.ascontiguousarray(white_state, dtype=np.float64)
        ys = np.ascontiguousarray(yolk_state, dtype=np.float64)
        out = C.c_int64()
        self._check(self._lib.egg_import_batch(self._h, C.byref(c), ws.ctypes.data, ys.ctypes.data, C.byref(out)))
        return out.value

    def remove(self, batch_id):  # L:140-155
        _assert_types(batch_id, "number")
        rc = self._check(self._lib.egg_remove(self._h, int(batch_id)))
        if rc == _ffi.EGG_OK:
            self._batch_colors.pop(int(batch_id), None)

    def draw(self, screen_size=(800, 600), origin=(0.0, 0.0), interpolation_alpha=None, clear=(0.0, 0.0, 0.0, 0.0),
             canvas_sizes=None, use_instancing=True):  # L:159-162
        """`draw()` without a window: _update_canvases + _draw_canvases (L:1995-2175) as HIP kernels into a float32
        RGBA image of screen_size = (width, height); world px = screen px + origin.  Returns an (H, W, 4) array.
        canvas_sizes = [(w, h) white, (w, h) yolk] overrides the sizes resize_canvas_maybe would pick (L:1935-1975)."""
        p = _ffi.EggRenderParams()
        self._check(self._lib.egg_default_render_params(C.byref(p)))
        p.screen_w, p.screen_h = int(screen_size[0]), int(screen_size[1])
        p.origin_x, p.origin_y = float(origin[0]), float(origin[1])
        if interpolation_alpha is not None:
            p.interpolation_alpha = float(interpolation_alpha)
        p.threshold = float(self._thresholding_threshold)
        p.smoothness = float(self._thresholding_smoothness)
        p.use_instancing = int(bool(use_instancing))
        if canvas_sizes is not None:
            for which in range(2):
                p.canvas_w[which], p.canvas_h[which] = int(canvas_sizes[which][0]), int(canvas_sizes[which][1])
        p.clear[:] = [float(c) for c in clear]
        # (the render config is NOT re-sent here: egg_set_render_config means "set_*_config was called" -- a new colour
        # table, L:1307-1311 -- and would end the sharing of the old one between the config and its colourless batches)
        image = np.empty((p.screen_h, p.screen_w, 4), dtype=np.float32)
        self._check(self._lib.egg_render(self._h, C.byref(p), image.ctypes.data_as(C.c_void_p)))
        return image

    def render_canvas(self, which):
        """the density canvas of `which` as the last draw() left it: ((h, w, 4) float32 array, (x0, y0) in world px)"""
        w, h, x0, y0 = C.c_int32(), C.c_int32(), C.c_double(), C.c_double()
        self._check(self._lib.egg_render_canvas(self._h, int(which), None, 0, C.byref(w), C.byref(h), C.byref(x0), C.byref(y0)))
        canvas = np.empty((h.value, w.value, 4), dtype=np.float32)
        self._check(self._lib.egg_render_canvas(self._h, int(which), canvas.ctypes.data_as(C.c_void_p), w.value * h.value,
                                                None, None, None, None))
        return canvas, (x0.value, y0.value)

    def particle_texture(self):
        """alpha of the particle density texture (L:620-682) the splat pass samples"""
        n = C.c_int32()
        self._check(self._lib.egg_render_particle_texture(self._h, None, 0, C.byref(n)))
        tex = np.empty((n.value, n.value), dtype=np.float32)
        self._check(self._lib.egg_render_particle_texture(self._h, tex.ctypes.data_as(C.c_void_p), tex.size, None))
        return tex

    # ------------------------------------------------------------------ update
    def update(self, delta, step_delta=None, n_substeps=None, n_collision_steps=None):  # L:168-222
        if step_delta is None:
            step_delta = 1 / 60
        if n_substeps is None:
            n_substeps = 2
        if n_collision_steps is None:
            n_collision_steps = 3
        _assert_types(delta, "number", step_delta, "number", n_substeps, "number", n_collision_steps, "number")
        if _is_nan(n_substeps) or _is_nan(n_collision_steps):
            raise EggError("[ERROR] In SimulationHandler.update: `n_substeps` is not a number > 0")
        n_substeps = math.ceil(n_substeps)  # L:181-182
        n_collision_steps = math.ceil(n_collision_steps)
        n = C.c_int32()
        self._check(self._lib.egg_update(self._h, float(delta), float(step_delta), int(n_substeps),
                                         int(n_collision_steps), C.byref(n)))
        return n.value

    def step(self, delta=1 / 60, n_substeps=2, n_collision_steps=3):
        """`_step` directly (L:1722); not part of the reference's public surface."""
        self._check(self._lib.egg_step(self._h, float(delta), int(n_substeps), int(n_collision_steps)))

    def step_begin(self, delta=1 / 60, n_substeps=2, n_collision_steps=3):
        """launch a `_step` without waiting for it (see egg_step_begin)"""
        self._check(self._lib.egg_step_begin(self._h, float(delta), int(n_substeps), int(n_collision_steps)))

    def step_end(self, commit=True):
        self._check(self._lib.egg_step_end(self._h, 1 if commit else 0))

    def step_peek_visits(self):
        """(max visits in one pass per type, budget per type) of the step launched with step_begin, before it is
        committed (see egg_step_peek_visits)"""
        v, b = (C.c_int64 * 2)(), (C.c_double * 2)()
        self._check(self._lib.egg_step_peek_visits(self._h, C.byref(v), C.byref(b)))
        return list(v), list(b)

    def prepare_step(self, step_delta=1 / 60, n_substeps=2, n_collision_steps=3):
        """form the tiles/claims of the next step without running it (multi-GPU exchange)"""
        self._check(self._lib.egg_prepare_step(self._h, float(step_delta), int(n_substeps), int(n_collision_steps)))

    # ---------------------------------------------------------------- targets
    def set_target_position(self, batch_id, x, y):  # L:254-264
        _assert_types(batch_id, "number", x, "number", y, "number")
        self._check(self._lib.egg_set_target(self._h, int(batch_id), float(x), float(y)))

    def set_target_positions(self, ids, xs, ys):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        xs = np.ascontiguousarray(xs, dtype=np.float64)
        ys = np.ascontiguousarray(ys, dtype=np.float64)
        self._check(self._lib.egg_set_targets_many(self._h, ids.shape[0], ids.ctypes.data, xs.ctypes.data,
                                                   ys.ctypes.data))

    def get_target_position(self, batch_id):  # L:268-278
        _assert_types(batch_id, "number")
        x, y = C.c_double(), C.c_double()
        self._check(self._lib.egg_get_target(self._h, int(batch_id), C.byref(x), C.byref(y)))
        return x.value, y.value

    def get_position(self, batch_id):  # L:281-295
        _assert_types(batch_id, "number")
        x, y = C.c_double(), C.c_double()
        self._check(self._lib.egg_get_position(self._h, int(batch_id), C.byref(x), C.byref(y)))
        return x.value, y.value

    def get_positions(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        xs = np.empty(ids.shape[0], dtype=np.float64)
        ys = np.empty(ids.shape[0], dtype=np.float64)
        self._check(self._lib.egg_get_positions_many(self._h, ids.shape[0], ids.ctypes.data, xs.ctypes.data,
                                                     ys.ctypes.data))
        return xs, ys

    def get_bounds(self, ids):
        """[n, 4] array (lo_x, lo_y, hi_x, hi_y) in px: the cells each batch's particles occupy."""
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        out = np.empty((4, ids.shape[0]), dtype=np.float64)
        self._check(self._lib.egg_get_bounds_many(self._h, ids.shape[0], ids.ctypes.data, out[0].ctypes.data,
                                                  out[1].ctypes.data, out[2].ctypes.data, out[3].ctypes.data))
        return out.T.copy()

    def get_claims(self, ids):
        """([n, 8] array: white box then yolk box, lo_x lo_y hi_x hi_y in px; (white cell, yolk cell))"""
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        out = np.empty((ids.shape[0], 8), dtype=np.float64)
        cells = np.empty(2, dtype=np.float64)
        self._check(self._lib.egg_get_claims_many(self._h, ids.shape[0], ids.ctypes.data, out.ctypes.data, cells.ctypes.data))
        return out, (float(cells[0]), float(cells[1]))

    # ----------------------------------------------------------------- colors
    # render attributes: kept on the host, never read by the solver (L:297-395)
    def _set_color(self, scope, which, batch_id, r, g, b, a):
        if a is None:
            a = 1
        _assert_types(batch_id, "number")
        _assert_types(r, "number", g, "number", b, "number", a, "number")
        if any(c > 1 or c < 0 for c in (r, g, b, a)):
            warnings.warn("In SimulationHandler.%s: color component is outside of [0, 1]" % scope, EggWarning)
        rgba = [min(max(c, 0), 1) for c in (r, g, b, a)]
        if int(batch_id) not in self._batch_colors:
            warnings.warn("In SimulationHandler.%s: no batch with id `%s`" % (scope, batch_id), EggWarning)
            return
        # in place (L:349-350, L:386-387): a batch created without a colour shares the config's table
        table = self._batch_colors[int(batch_id)][which]
        if isinstance(table, list):
            table[:] = rgba
        else:
            self._batch_colors[int(batch_id)][which] = rgba
        # its particles (L:1110-1129) and, when the batch shares the config's table, the config's colour: the library
        # applies the same aliasing as the tables above
        self._lib.egg_set_color(self._h, int(batch_id), int(which), *[float(c) for c in rgba])

    def set_white_color(self, batch_id, r, g, b, a=None, *outline):  # L:365-394
        self._set_color("set_white_color", 0, batch_id, r, g, b, a)

    def set_yolk_color(self, batch_id, r, g, b, a=None, *outline):  # L:328-357
        self._set_color("set_egg_yolk_color", 1, batch_id, r, g, b, a)

    # ------------------------------------------------------------ bookkeeping
    def list_ids(self):  # L:399-405
        n = C.c_int64()
        self._check(self._lib.egg_list_ids(self._h, 0, None, C.byref(n)))
        ids = np.empty(n.value, dtype=np.int64)
        self._check(self._lib.egg_list_ids(self._h, n.value, ids.ctypes.data, C.byref(n)))
        return [int(i) for i in ids]

    def get_n_particles(self, batch_or_nil=None):  # L:409-419
        w, y = C.c_int64(), C.c_int64()
        self._check(self._lib.egg_get_n_particles(self._h, -1 if batch_or_nil is None else int(batch_or_nil),
                                                  C.byref(w), C.byref(y)))
        return w.value, y.value

    @property
    def elapsed(self):
        e, a = C.c_double(), C.c_double()
        self._lib.egg_get_elapsed(self._h, C.byref(e), C.byref(a))
        return e.value

    @property
    def interpolation_alpha(self):
        e, a = C.c_double(), C.c_double()
        self._lib.egg_get_elapsed(self._h, C.byref(e), C.byref(a))
        return a.value

    # ---------------------------------------------------------- device access
    def synchronize(self):
        self._check(self._lib.egg_synchronize(self._h))

    def download(self, which, field):
        """One particle field (name from _ffi.FIELDS) of every particle, particle-index order."""
        w, y = self.get_n_particles()
        n = w if which == _ffi.WHITE else y
        out = np.empty(n, dtype=np.float64)
        self._check(self._lib.egg_download_particles(self._h, which, _ffi.FIELD_ID[field], out.ctypes.data, n))
        return out

    def download_instance_data(self, which):
        """The reference's instanced-draw record per particle (L:513-517, L:744-813):
        columns x, y, last_x, last_y, vx, vy, radius."""
        cols = [self.download(which, f) for f in ("x", "y", "last_x", "last_y", "vx", "vy", "radius")]
        return np.stack(cols, axis=1) if cols[0].size else np.zeros((0, 7))

    def get_environment(self, which):
        """the reductions the reference keeps per particle type for :draw() -- AABB incl. radius, centroid, largest
        radius and speed, centroid at the start of the last step (simulation_handler.lua:1669-1718, 1795-1815)"""
        e = _ffi.EggEnvironment()
        self._check(self._lib.egg_get_environment(self._h, int(which), C.byref(e)))
        return {k: getattr(e, k) for k in _ffi.ENVIRONMENT_FIELDS}

    def stats(self):
        s = _ffi.EggStats()
        self._check(self._lib.egg_get_stats(self._h, C.byref(s)))
        return dict(steps=s.steps, pair_solves=s.pair_solves, follow_solves=s.follow_solves,
                    kernel_launches=s.kernel_launches, retiles=s.retiles, redo_steps=s.redo_steps,
                    n_tiles=list(s.n_tiles), max_tile_particles=list(s.max_tile_particles),
                    last_step_kernel_ms=s.last_step_kernel_ms, single_tile=list(s.single_tile),
                    kernel_ms=list(s.kernel_ms), kernel_ms_sum=list(s.kernel_ms_sum), timed_steps=s.timed_steps,
                    max_pass_visits=list(s.max_pass_visits), budget=list(s.budget), fused_launch=int(s.fused_launch),
                    packed=list(s.packed), pk_kernel_ms=[list(r) for r in s.pk_kernel_ms],
                    pk_kernel_launches=[list(r) for r in s.pk_kernel_launches], host_ms=list(s.host_ms), max_levels=list(s.max_levels),
                    pk_variants=list(s.pk_variants))

    def selftest_arith(self, n=1 << 24, seed=1):
        """mismatches of the kernel's hand-expanded f64 division against `/` on n random operand pairs"""
        bad = C.c_int64()
        self._check(self._lib.egg_selftest_arith(self._h, int(n), int(seed), C.byref(bad)))
        return bad.value

    def set_option(self, option, value):
        self._check(self._lib.egg_set_option(self._h, int(option), float(value)))
