"""Drop-in module name of the reference (`Cargo.toml:7`, `src/lib.rs:380`):
`from arrowspace import ArrowSpaceBuilder` resolves to the MI355X-native implementation."""
from pyarrowspace_amd import (ArrowSpace, ArrowSpaceBuilder, GraphLaplacian, PanicException,  # noqa: F401
                              set_debug)

__all__ = ["ArrowSpaceBuilder", "ArrowSpace", "GraphLaplacian", "set_debug", "PanicException"]
