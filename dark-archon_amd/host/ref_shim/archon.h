// ref_shim/archon.h -- lets the reference's own caller (kvark/dark-archon bwt/a7/src/main.cpp:5, `#include "archon.h"`)
// pick up the MI355X-backed class Archon instead of bwt/a7/src/archon.h:8-29.  See INTEGRATION.md, option A.
#include "../archon_host.h"
