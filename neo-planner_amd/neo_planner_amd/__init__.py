"""
neo_planner_amd -- MI355X-native replan inner loop of neo-planner (MINCO minimum-jerk
trajectory optimisation: coefficient solve, cost/gradient with ESDF lookups, L-BFGS-B),
behind the reference's Python interface.  See DESIGN.md / INTEGRATION.md.

Importing the package does not touch the GPU; the HIP library is loaded on first use and
there is no CPU fallback.
"""
from ._lib import Context, NeoError, default_context  # noqa: F401
from .esdf import ESDF, ESDF3D  # noqa: F401
from .planner import BatchPlanner, MinJerkPlanner, PlannerConfig  # noqa: F401
# the initializer network (torch) is imported on demand: `from neo_planner_amd import initializer`

__all__ = ["Context", "NeoError", "default_context", "ESDF", "ESDF3D", "BatchPlanner", "MinJerkPlanner",
           "PlannerConfig"]
