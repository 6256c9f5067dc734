"""Empty shell of the un-vendored SCNet repo's `utils` package (see ../README.md).  TEST INFRASTRUCTURE ONLY."""
