"""TEST STUB (tests/stubs/README.md)."""
