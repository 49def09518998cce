"""exahip — host-side Python mirror of the ExaModels evaluation surface over libexahip.so (MI355X)."""
from .core import ExaCore, Table, URange, product, rng
from .recipe import length  # noqa: F401
from . import graph, models  # noqa: F401
from .model import CompressedExaModel, ExaModel, Recipe, TimedExaModel  # noqa: F401
