--- Headless stand-in for the LOVE (love2d) globals the reference's SimulationHandler touches while it is
--- CONSTRUCTED and STEPPED (simulation_handler.lua:494-559 instancing probe and texture formats, :567-680 shaders
--- and particle texture, :1935-1978 canvas resize inside _step).  Nothing is rendered; every graphics object is
--- a table of no-ops.  Written against SURVEY.md 8c (last row); NOT EXECUTED in this pipeline (no Lua
--- interpreter in the image) -- it exists so that ONE LuaJIT run elsewhere can pin the oracle, see gen_golden.lua.
local function dummy_object()
    local o = {}
    local noop = function() end
    for _, name in ipairs({ "setFilter", "setWrap", "setVertexMap", "setTexture", "setVertexAttribute", "attachAttribute",
                            "setVertices", "setDrawRange", "send", "release", "flush" }) do
        o[name] = noop
    end
    o.getDimensions = function() return 1, 1 end
    o.getWidth = function() return 1 end
    o.getHeight = function() return 1 end
    o.hasUniform = function() return false end
    return o
end

local noop = function() end
love = {
    getVersion = function() return 12, 0, 0 end,
    filesystem = { getInfo = function() return {} end },
    math = { random = math.random },
    timer = { getTime = os.clock },
    graphics = {
        getSupported = function() return {} end,                 -- no instancing: _use_instancing = false (:494-496)
        getTextureFormats = function() return { rgba8 = true } end,
        getRendererInfo = function() return "OpenGL" end,
        validateShader = function() return true end,
        newShader = dummy_object, newCanvas = dummy_object, newMesh = dummy_object,
        push = noop, pop = noop, reset = noop, clear = noop, setCanvas = noop, setShader = noop, setBlendMode = noop,
        translate = noop, scale = noop, setColor = noop, draw = noop, drawInstanced = noop, rectangle = noop,
    },
}
return love
