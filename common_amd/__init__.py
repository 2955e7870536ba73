"""common_amd -- MI355X (gfx950) implementation of the component-model scoring hot
path of datamicroscopes/common (score_value / score_data / add_value /
remove_value per row x group x feature), behind the reference's model/hypers/group
surface.  See DESIGN.md; the C ABI is include/microscopes_hip.h.

Importing the package does not touch the GPU; creating a Context does, and fails
loudly when the HIP library or a gfx950 device is missing (there is no CPU path).
"""
from ._lib import (ACC_NO_COMMIT, ACC_RESET, ACC_SUBTRACT, BB, BBNC, BNB, DD, DM, GP, NICH, NIW, NOOP,
                   SCORE_CRP_PRIOR, MicroscopesHipError, EXPORTS, LIB_PATH, load)
from .runtime import Context, DataView, RelationView, SparseRelationView, State, pack_hp, runtime_types_of, ss_dtype, type_of_numpy
from . import dist, models

__all__ = ["Context", "DataView", "RelationView", "SparseRelationView", "State", "BNB", "DM", "models", "dist", "load", "MicroscopesHipError", "BB", "BBNC", "GP", "DD",
           "NICH", "NIW", "NOOP", "pack_hp", "ss_dtype", "runtime_types_of", "type_of_numpy",
           "EXPORTS", "LIB_PATH", "SCORE_CRP_PRIOR", "ACC_RESET", "ACC_SUBTRACT", "ACC_NO_COMMIT"]
