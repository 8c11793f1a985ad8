"""TEST INFRASTRUCTURE ONLY -- CPU restatements of the reference algorithms.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import from here, and only as the checker / the timed CPU baseline.  The
shipped path (``microimagelib_amd`` + ``libapi.so``) never imports it.
"""
