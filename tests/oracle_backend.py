"""tests/oracle_backend.py -- shim: the CPU stand-ins live in oracle/backend.py (test infrastructure)."""
from oracle.backend import *  # noqa: F401,F403
from oracle.backend import install  # noqa: F401
