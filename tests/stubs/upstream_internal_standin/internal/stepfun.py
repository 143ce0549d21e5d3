"""TEST STAND-IN (see configs.py next to this file): only its presence matters."""
