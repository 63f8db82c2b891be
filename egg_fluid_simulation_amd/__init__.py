"""egg_fluid_simulation_amd -- MI355X (gfx950) implementation of the XPBD particle
step of Clemapfel/egg_fluid_simulation behind the reference's SimulationHandler API.

    from egg_fluid_simulation_amd import SimulationHandler
    handler = SimulationHandler()            # default white / yolk configs
    egg = handler.add(400, 300, 50, 15)
    handler.set_target_position(egg, 450, 300)
    handler.update(1 / 60)
    x, y = handler.get_position(egg)

The solver runs only on the GPU through libeggsim.so (include/eggsim.h); importing
this package never imports the CPU oracle under oracle/.
"""
from . import _ffi
from .default_config import default_configs
from .simulation_handler import EggError, EggWarning, SimulationHandler
from .group import SimulationGroup

WHITE, YOLK = _ffi.WHITE, _ffi.YOLK

__all__ = ["SimulationHandler", "SimulationGroup", "EggError", "EggWarning", "default_configs", "WHITE", "YOLK"]
