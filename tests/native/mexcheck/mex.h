/* COMPILE CHECK ONLY: lets `#include "mex.h"` in matlab/ *.c resolve to the declarations-only tests/native/mex_decls.h
 * under `gcc -fsyntax-only` (tests/test_abi_and_host.py).  Never on any build's or the oracle's include path. */
#include "../mex_decls.h"
