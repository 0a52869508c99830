"""ORACLE — test infrastructure only (see oracle/MinkowskiEngine/__init__.py)."""
