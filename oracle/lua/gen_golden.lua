--- Dumps reference-held golden vectors: the cases of oracle/gen_golden.py, run by the REFERENCE ITSELF.
---
---   luajit oracle/lua/gen_golden.lua <dir that CONTAINS egg_fluid_simulation/> <output dir>
---
--- <dir>/egg_fluid_simulation/ must be a checkout of Clemapfel/egg_fluid_simulation (the directory name is fixed by
--- simulation_handler.lua:1,6-7).  The reference is required in place, nothing of it is copied.  Output: one text
--- file per case, lines `name count v1 v2 ...` with every double printed as %.17g (round-trips exactly);
--- oracle/lua/import_golden.py turns them into tests/golden_lua/*.npz, which tests/test_oracle.py then holds the C
--- oracle (and through it the device path) to, bit for bit.  NOT EXECUTED in this pipeline: no Lua interpreter
--- exists in the build or GPU image (SURVEY.md 8c); until someone runs it, parity stays "unpinned".
local root, out_dir = arg[1], arg[2]
assert(root and out_dir, "usage: luajit gen_golden.lua <dir containing egg_fluid_simulation/> <output dir>")
package.path = root .. "/?.lua;" .. root .. "/?/init.lua;" .. package.path
dofile((arg[0]:match("^(.*)/[^/]*$") or ".") .. "/love_stub.lua")
local SimulationHandler = require("egg_fluid_simulation.simulation_handler")
-- the config file returns two tables (white, yolk): load it the way the reference's demo does (test.lua:10-13)
local default_white, default_yolk = assert(loadfile(root .. "/egg_fluid_simulation/simulation_handler_default_config.lua"))()

local STRIDE = 20 -- simulation_handler.lua:713-735
local X, Y, VX, VY, RADIUS, MASS_T, INV_MASS = 0, 1, 3, 4, 7, 8, 10

local CASES = { -- name, centers, moving target, steps, snapshot steps, sub-steps, collision passes (oracle/gen_golden.py)
    { "cfg1_static", { { 400, 300 } }, false, 100, { 1, 2, 10, 100 }, 2, 3 },
    { "cfg1_moving", { { 400, 300 } }, true, 100, { 1, 2, 10, 100 }, 2, 3 },
    { "cfg1_origin", { { 0, 0 } }, true, 50, { 1, 10, 50 }, 2, 3 },
    { "four_batches", { { 0, 0 }, { 30, 10 }, { -20, 40 }, { 200, 200 } }, true, 20, { 1, 5, 20 }, 2, 3 },
    { "substeps_3_2", { { 10, 10 }, { 60, 10 } }, true, 12, { 1, 12 }, 3, 2 },
    { "substeps_2_1", { { 10, 10 }, { 20, 20 } }, true, 12, { 1, 12 }, 2, 1 },
}

local function field(data, offset)
    local out = {}
    for i = 1, #data / STRIDE do out[i] = data[(i - 1) * STRIDE + 1 + offset] end
    return out
end

local function emit(file, name, values)
    local parts = { name, tostring(#values) }
    for i, v in ipairs(values) do parts[i + 2] = string.format("%.17g", v) end
    file:write(table.concat(parts, " "), "\n")
end

for _, case in ipairs(CASES) do
    local name, centers, moving, n_steps, snaps, S, C = unpack(case)
    local handler = SimulationHandler(default_white, default_yolk)
    local ids = {}
    for i, c in ipairs(centers) do ids[i] = handler:add(c[1], c[2], 50, 15) end
    local file = assert(io.open(out_dir .. "/" .. name .. ".txt", "w"))
    for tag, data in pairs({ white = handler._white_data, yolk = handler._yolk_data }) do
        for fname, off in pairs({ x = X, y = Y, mass_t = MASS_T, inv_mass = INV_MASS, radius = RADIUS }) do
            emit(file, "init_" .. tag .. "_" .. fname, field(data, off))
        end
    end
    local is_snap = {}
    for _, s in ipairs(snaps) do is_snap[s] = true end
    for k = 0, n_steps - 1 do
        if moving then -- gate-B trajectory: 100 px circle, one revolution per 100 steps
            for i, c in ipairs(centers) do
                handler:set_target_position(ids[i], c[1] + 100 * math.cos(2 * math.pi * k / 100),
                    c[2] + 100 * math.sin(2 * math.pi * k / 100))
            end
        end
        handler:update(1 / 60, 1 / 60, S, C)
        if is_snap[k + 1] then
            for tag, data in pairs({ white = handler._white_data, yolk = handler._yolk_data }) do
                for fname, off in pairs({ x = X, y = Y, vx = VX, vy = VY }) do
                    emit(file, tag .. "_step" .. (k + 1) .. "_" .. fname, field(data, off))
                end
            end
            local cen = {}
            for i, id in ipairs(ids) do
                local cx, cy = handler:get_position(id)
                cen[2 * i - 1], cen[2 * i] = cx, cy
            end
            emit(file, "centroid_step" .. (k + 1), cen)
        end
    end
    file:close()
    print("wrote " .. name)
end
