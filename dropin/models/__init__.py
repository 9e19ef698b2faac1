"""Stands in for the reference's un-vendored `models` submodule (.gitmodules:1-3): a regular package, so it also wins
over an empty `models/` directory left by an un-initialised submodule."""
