// stubs.hip -- (empty) every entry point of include/julius_amd.h now has a device
// implementation; kept so the Makefile's wildcard has a stable file list.
#include "jamd_internal.h"
