--- Drop-in replacement for egg_fluid_simulation/simulation_handler.lua whose particle
--- solver runs on an AMD MI355X through libeggsim.so (include/eggsim.h).
---
--- NOT EXECUTED IN THIS PIPELINE: no Lua / LuaJIT interpreter exists in the build or GPU image
--- (SURVEY.md 8c).  The file is kept thin and declarative; every FFI call it makes is mirrored
--- one-to-one by egg_fluid_simulation_amd/simulation_handler.py over ctypes, which IS tested.
---
--- Surface kept from the reference (same names, argument order, defaults, warnings, errors):
---   SimulationHandler(white_config, yolk_config), :add, :remove, :update, :set_target_position,
---   :get_target_position, :get_position, :set_white_config/:set_yolk_config, :get_*_config,
---   :list_ids, :get_n_particles, :set_white_color, :set_yolk_color.  In a LOVE host :draw stays with the reference's
---   own shaders: :get_instance_data() returns the instanced-draw record (x, y, last_x, last_y, vx, vy, radius) and
---   :get_environment() what sizes and places the canvases.  Without a window, :render_to_image() runs the same
---   passes as HIP kernels (include/eggsim.h, "headless renderer") into a float32 RGBA buffer.

local prefix = "egg_fluid_simulation"
require(string.gsub(prefix .. "/math", "[/\\]", "."))
local log = require(string.gsub(prefix .. "/log", "[/\\]", "."))
local ffi = require("ffi")

ffi.cdef[[
typedef struct egg_handle egg_handle;
typedef struct {
    double damping, follow_strength, cohesion_strength, cohesion_interaction_distance_factor,
           collision_strength, collision_overlap_factor, min_mass, max_mass, min_radius, max_radius,
           max_collision_fraction, mass_distribution_variance, eps;
} egg_config;
int egg_create(const egg_config *white, const egg_config *yolk, int device, egg_handle **out);
void egg_destroy(egg_handle *h);
const char *egg_last_error(const egg_handle *h);
int egg_set_config(egg_handle *h, int which, const egg_config *cfg);
int egg_add(egg_handle *h, double x, double y, double white_radius, double yolk_radius,
            int64_t white_n, int64_t yolk_n, int64_t *out_id);
int egg_remove(egg_handle *h, int64_t id);
int egg_set_target(egg_handle *h, int64_t id, double x, double y);
int egg_get_target(const egg_handle *h, int64_t id, double *x, double *y);
int egg_update(egg_handle *h, double delta, double step_delta, int32_t n_substeps,
               int32_t n_collision_steps, int32_t *out_n_steps);
int egg_get_position(egg_handle *h, int64_t id, double *x, double *y);
int egg_get_n_particles(const egg_handle *h, int64_t id, int64_t *n_white, int64_t *n_yolk);
int egg_list_ids(const egg_handle *h, int64_t cap, int64_t *ids, int64_t *n);
int egg_download_particles(egg_handle *h, int which, int field, double *dst, int64_t cap);
typedef struct { double min_x, min_y, max_x, max_y, centroid_x, centroid_y, max_radius, max_velocity,
                 last_centroid_x, last_centroid_y; } egg_environment;
int egg_get_environment(egg_handle *h, int which, egg_environment *out);
typedef struct {
    float color[4], outline_color[4];
    double outline_thickness;
    double highlight_strength, shadow_strength;
    double texture_scale, motion_blur;
} egg_render_config;
int egg_set_render_config(egg_handle *h, int which, const egg_render_config *cfg);
int egg_set_add_color(egg_handle *h, int64_t id, int which, double r, double g, double b, double a);
int egg_set_color(egg_handle *h, int64_t id, int which, double r, double g, double b, double a);
typedef struct {
    int32_t screen_w, screen_h;
    double origin_x, origin_y;
    double interpolation_alpha;
    double threshold, smoothness;
    int32_t use_instancing;
    int32_t canvas_w[2], canvas_h[2];
    float clear[4];
} egg_render_params;
int egg_default_render_params(egg_render_params *p);
int egg_render(egg_handle *h, const egg_render_params *p, float *rgba);
]]

local lib = ffi.load(os.getenv("EGGSIM_LIB") or "eggsim")
local NaN = 0 / 0
local EGG_DEFAULT_COUNT = -1 -- include/eggsim.h

local SimulationHandler = {}
setmetatable(SimulationHandler, { __call = function(_, ...) return SimulationHandler.new(...) end })
local _type_metatable = { __index = SimulationHandler }

-- keys, bounds and messages of the reference's _load_config (simulation_handler.lua:1152-1320)
local _valid_config_keys = {
    damping = { type = "number", min = 0, max = 1 }, color = { type = "color" }, outline_color = { type = "color" },
    outline_thickness = { type = "number", min = 0 }, collision_strength = { type = "number", min = 0, max = 1 },
    collision_overlap_factor = { type = "number", min = 0 }, cohesion_strength = { type = "number", min = 0, max = 1 },
    cohesion_interaction_distance_factor = { type = "number", min = 0 }, follow_strength = { type = "number", min = 0, max = 1 },
    min_radius = { type = "number", min = 0 }, max_radius = { type = "number", min = 0 },
    min_mass = { type = "number", min = 0 }, max_mass = { type = "number", min = 0 },
    motion_blur = { type = "number", min = 0, max = 1 }, texture_scale = { type = "number", min = 1 },
    highlight_strength = { type = "number", min = 0 }, shadow_strength = { type = "number", min = 0 },
}
local _solver_keys = { "damping", "follow_strength", "cohesion_strength", "cohesion_interaction_distance_factor",
    "collision_strength", "collision_overlap_factor", "min_mass", "max_mass", "min_radius", "max_radius" }

local function _deepcopy(v)
    if type(v) ~= "table" then return v end
    local out = {}
    for k, x in pairs(v) do out[k] = _deepcopy(x) end
    return out
end

--- status -> the reference's conventions: > 0 warns and carries on, < 0 throws (log.lua:9-46)
function SimulationHandler:_check(rc)
    if rc == 0 then return rc end
    local message = ffi.string(lib.egg_last_error(self._h))
    if rc > 0 then log.warning(message) else log.error(message) end
    return rc
end

function SimulationHandler:_load_config(config, white_or_yolk)
    local scope = white_or_yolk and "In SimulationHandler.set_white_config: " or "In SimulationHandler.set_yolk_config: "
    local target = white_or_yolk and self._white_config or self._yolk_config
    for key, value in pairs(config) do
        local entry = _valid_config_keys[key]
        if entry == nil then
            log.warning(scope, "unrecognized config key `", key, "`, it will be ignored")
        elseif entry.type == "color" then
            for i = 1, 4 do
                local c = value[i]
                if c == nil or #value > 4 then log.error(scope, "color `", key, "` does not have 4 components") return end
                if type(c) ~= "number" or math.is_nan(c) then log.error(scope, "color `", key, "` has a component that is not a number") return end
                if c < 0 or c > 1 then log.warning(scope, "color `", key, "` has a component that is outside of [0, 1]") end
                value[i] = math.clamp(c, 0, 1)
            end
            target[key] = value
        elseif type(value) ~= entry.type then
            log.error(scope, "wrong type for config key `", key, "`, expected `", entry.type, "`, got `", type(value), "`")
            return
        elseif math.is_nan(value) then
            log.warning(scope, "config key `", key, "` is NaN, it will be ignored")
        else
            if entry.min ~= nil and value < entry.min then
                log.warning(scope, "config key `", key, "`'s value is `", value, "`, expected a value larger than `", entry.min, "`")
                value = math.max(value, entry.min)
            elseif entry.max ~= nil and value > entry.max then
                log.warning(scope, "config key `", key, "`'s value is `", value, "`, expected a value smaller than `", entry.max, "`")
                value = math.min(value, entry.max)
            end
            target[key] = value
        end
    end
end

function SimulationHandler:_c_config(white_or_yolk)
    local cfg = white_or_yolk and self._white_config or self._yolk_config
    local c = ffi.new("egg_config")
    for _, key in ipairs(_solver_keys) do c[key] = cfg[key] end
    c.max_collision_fraction = self._max_collision_fraction
    c.mass_distribution_variance = self._mass_distribution_variance
    c.eps = math.eps
    return c
end

function SimulationHandler.new(white_config, yolk_config, device)
    if yolk_config == nil then yolk_config = white_config end
    log.assert(white_config, "table", yolk_config, "table")
    local self = setmetatable({}, _type_metatable)
    self._white_config, self._yolk_config = {}, {}
    self._mass_distribution_variance = 4
    self._max_collision_fraction = 0.05
    self._batch_colors = {} -- render attribute, host side only (simulation_handler.lua:297-395)
    self:_load_config(_deepcopy(white_config), true)
    self:_load_config(_deepcopy(yolk_config), false)
    local out = ffi.new("egg_handle*[1]")
    local rc = lib.egg_create(self:_c_config(true), self:_c_config(false), device or 0, out)
    if rc ~= 0 then log.error("In SimulationHandler.new: ", ffi.string(lib.egg_last_error(nil))) end
    self._h = ffi.gc(out[0], lib.egg_destroy)
    return self
end

function SimulationHandler:add(x, y, white_radius, yolk_radius, white_color, yolk_color, white_n_particles, yolk_n_particles)
    -- argument handling of the reference's add (simulation_handler.lua:27-120), in its order: defaults, type
    -- assertion, radius / count errors, colour errors and warnings; the particle-count defaults themselves
    -- are computed by the library (EGG_DEFAULT_COUNT = "the caller gave nil"), from the same formula (L:52-58)
    -- (whether the caller gave a colour decides if the batch gets a table of its own or shares the config's, L:49-50)
    local given_white, given_yolk = white_color ~= nil, yolk_color ~= nil
    white_color = white_color or self._white_config.color
    yolk_color = yolk_color or self._yolk_config.color
    log.assert(x, "number", y, "number")
    if white_radius ~= nil then log.assert(white_radius, "number") end
    if yolk_radius ~= nil then log.assert(yolk_radius, "number") end
    log.assert(white_color, "table", yolk_color, "table")
    if white_n_particles ~= nil then log.assert(white_n_particles, "number") end
    if yolk_n_particles ~= nil then log.assert(yolk_n_particles, "number") end

    if white_radius ~= nil and white_radius <= 0 then
        log.error("In SimulationHandler.add: white radius cannot be 0 or negative")
    end
    if yolk_radius ~= nil and yolk_radius <= 0 then
        log.error("In SimulationHandler.add: yolk radius cannot be 0 or negative")
    end
    if white_n_particles ~= nil and white_n_particles <= 1 then
        log.error("In SimulationHandler.add: white particle count cannot be 1 or negative")
    end
    if yolk_n_particles ~= nil and yolk_n_particles <= 1 then
        log.error("In SimulationHandler.add: yolk particle count cannot be 1 or negative")
    end

    local component_names = { "r", "g", "b", "a" }
    for _, entry in ipairs({ { "white", white_color }, { "yolk", yolk_color } }) do
        local name, color = entry[1], entry[2]
        for i, component_name in ipairs(component_names) do
            if type(color[i]) ~= "number" or math.is_nan(color[i]) then
                log.error("In SimulationHandler.add: ", name, " color component `", component_name, "` is not a number")
                return
            end
            if color[i] < 0 or color[i] > 1 then
                log.warning("In SimulationHandler.add: ", name, " color component `", component_name, "` is outside of [0, 1]")
            end
            color[i] = math.clamp(color[i], 0, 1)
        end
    end

    local id = ffi.new("int64_t[1]")
    -- status 2 (EGG_WARN_FEW_PARTICLES) carries the reference's "only `n` particles will be created" warning
    self:_check(lib.egg_add(self._h, x, y, white_radius or NaN, yolk_radius or NaN,
        white_n_particles and math.ceil(white_n_particles) or EGG_DEFAULT_COUNT,
        yolk_n_particles and math.ceil(yolk_n_particles) or EGG_DEFAULT_COUNT, id))
    local batch_id = tonumber(id[0])
    -- a batch created without a colour shares the config's colour table (simulation_handler.lua:49-50)
    self._batch_colors[batch_id] = { white_color, yolk_color }
    if given_white then lib.egg_set_add_color(self._h, batch_id, 0, white_color[1], white_color[2], white_color[3], white_color[4]) end
    if given_yolk then lib.egg_set_add_color(self._h, batch_id, 1, yolk_color[1], yolk_color[2], yolk_color[3], yolk_color[4]) end
    return batch_id
end

--- the render keys of a config table as egg_render_config (simulation_handler_default_config.lua:22-36)
function SimulationHandler:_c_render_config(white_or_yolk)
    local cfg = white_or_yolk and self._white_config or self._yolk_config
    local c = ffi.new("egg_render_config")
    for i = 1, 4 do
        c.color[i - 1] = cfg.color[i]
        c.outline_color[i - 1] = cfg.outline_color[i]
    end
    c.outline_thickness, c.highlight_strength, c.shadow_strength = cfg.outline_thickness, cfg.highlight_strength, cfg.shadow_strength
    c.texture_scale, c.motion_blur = cfg.texture_scale, cfg.motion_blur
    return c
end

--- set_white_color / set_yolk_color (simulation_handler.lua:328-398): in place, like the reference
function SimulationHandler:_set_color(scope, which, batch_id, r, g, b, a)
    if a == nil then a = 1 end
    log.assert(batch_id, "number", r, "number", g, "number", b, "number", a, "number")
    if r > 1 or r < 0 or g > 1 or g < 0 or b > 1 or b < 0 or a > 1 or a < 0 then
        log.warning("In SimulationHandler.", scope, ": color component is outside of [0, 1]")
    end
    local colors = self._batch_colors[batch_id]
    if colors == nil then
        log.warning("In SimulationHandler.", scope, ": no batch with id `", batch_id, "`")
        return
    end
    local color = colors[which + 1]
    color[1], color[2], color[3], color[4] = math.clamp(r, 0, 1), math.clamp(g, 0, 1), math.clamp(b, 0, 1), math.clamp(a, 0, 1)
    lib.egg_set_color(self._h, batch_id, which, color[1], color[2], color[3], color[4])
end
function SimulationHandler:set_white_color(batch_id, r, g, b, a) self:_set_color("set_white_color", 0, batch_id, r, g, b, a) end
function SimulationHandler:set_yolk_color(batch_id, r, g, b, a) self:_set_color("set_egg_yolk_color", 1, batch_id, r, g, b, a) end

--- :draw() without a window: both passes of the reference's draw path as HIP kernels (include/eggsim.h) into a
--- float32 RGBA buffer of width x height; world px = screen px + origin.  Returns the buffer (row-major).
function SimulationHandler:render_to_image(width, height, origin_x, origin_y)
    local p = ffi.new("egg_render_params[1]")
    lib.egg_default_render_params(p)
    p[0].screen_w, p[0].screen_h = width, height
    p[0].origin_x, p[0].origin_y = origin_x or 0, origin_y or 0
    for which = 0, 1 do self:_check(lib.egg_set_render_config(self._h, which, self:_c_render_config(which == 0))) end
    local image = ffi.new("float[?]", width * height * 4)
    self:_check(lib.egg_render(self._h, p, image))
    return image
end

function SimulationHandler:remove(batch_id)
    log.assert(batch_id, "number")
    if self:_check(lib.egg_remove(self._h, batch_id)) == 0 then self._batch_colors[batch_id] = nil end
end

function SimulationHandler:update(delta, step_delta, n_substeps, n_collision_steps)
    if step_delta == nil then step_delta = 1 / 60 end
    if n_substeps == nil then n_substeps = 2 end
    if n_collision_steps == nil then n_collision_steps = 3 end
    log.assert(delta, "number", step_delta, "number", n_substeps, "number", n_collision_steps, "number")
    local n = ffi.new("int32_t[1]")
    self:_check(lib.egg_update(self._h, delta, step_delta, math.ceil(n_substeps), math.ceil(n_collision_steps), n))
    return n[0]
end

function SimulationHandler:set_white_config(config)
    log.assert(config, "table")
    self:_load_config(_deepcopy(config), true)
    self:_check(lib.egg_set_config(self._h, 0, self:_c_config(true)))
end

function SimulationHandler:set_yolk_config(config)
    log.assert(config, "table")
    self:_load_config(_deepcopy(config), false)
    self:_check(lib.egg_set_config(self._h, 1, self:_c_config(false)))
end

function SimulationHandler:get_white_config() return _deepcopy(self._white_config) end
function SimulationHandler:get_yolk_config() return _deepcopy(self._yolk_config) end

function SimulationHandler:set_target_position(batch_id, x, y)
    log.assert(batch_id, "number", x, "number", y, "number")
    self:_check(lib.egg_set_target(self._h, batch_id, x, y))
end

function SimulationHandler:get_target_position(batch_id)
    log.assert(batch_id, "number")
    local x, y = ffi.new("double[1]"), ffi.new("double[1]")
    if self:_check(lib.egg_get_target(self._h, batch_id, x, y)) ~= 0 then return nil, nil end
    return x[0], y[0]
end

function SimulationHandler:get_position(batch_id)
    log.assert(batch_id, "number")
    local x, y = ffi.new("double[1]"), ffi.new("double[1]")
    if self:_check(lib.egg_get_position(self._h, batch_id, x, y)) ~= 0 then return nil, nil end
    return x[0], y[0]
end

function SimulationHandler:list_ids()
    local n = ffi.new("int64_t[1]")
    self:_check(lib.egg_list_ids(self._h, 0, nil, n))
    local buf = ffi.new("int64_t[?]", math.max(1, tonumber(n[0])))
    self:_check(lib.egg_list_ids(self._h, n[0], buf, n))
    local ids = {}
    for i = 0, tonumber(n[0]) - 1 do ids[i + 1] = tonumber(buf[i]) end
    return ids
end

function SimulationHandler:get_n_particles(batch_or_nil)
    local w, y = ffi.new("int64_t[1]"), ffi.new("int64_t[1]")
    self:_check(lib.egg_get_n_particles(self._h, batch_or_nil == nil and -1 or batch_or_nil, w, y))
    return tonumber(w[0]), tonumber(y[0])
end

--- the reference's instanced-draw record per particle (simulation_handler.lua:513-517, 744-813)
function SimulationHandler:get_instance_data(white_or_yolk)
    local which = white_or_yolk and 0 or 1
    local n_white, n_yolk = self:get_n_particles()
    local n = white_or_yolk and n_white or n_yolk
    local fields = { 0, 1, 4, 5, 2, 3, 6 } -- x, y, last_x, last_y, vx, vy, radius
    local out = {}
    for col, field in ipairs(fields) do
        local buf = ffi.new("double[?]", math.max(1, n))
        self:_check(lib.egg_download_particles(self._h, which, field, buf, n))
        out[col] = buf
    end
    return out, n
end

--- the fields the reference's environments hold for :draw(): particle AABB incl. radius, centroid, largest
--- radius / speed, centroid at the start of the last step (simulation_handler.lua:1669-1718, 1795-1815,
--- used at 1946-1950, 2007, 2132 to size and place the canvases)
function SimulationHandler:get_environment(white_or_yolk)
    local e = ffi.new("egg_environment[1]")
    self:_check(lib.egg_get_environment(self._h, white_or_yolk and 0 or 1, e))
    local v = e[0]
    return { min_x = v.min_x, min_y = v.min_y, max_x = v.max_x, max_y = v.max_y, centroid_x = v.centroid_x,
             centroid_y = v.centroid_y, max_radius = v.max_radius, max_velocity = v.max_velocity,
             last_centroid_x = v.last_centroid_x, last_centroid_y = v.last_centroid_y }
end

function SimulationHandler:draw()
    -- in a LOVE host: feed :get_instance_data() and :get_environment() to the reference's shaders and canvas code
    -- (simulation_handler.lua:1995-2175); without a window use :render_to_image()
end

return SimulationHandler
