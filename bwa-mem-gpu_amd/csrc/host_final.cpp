// Host-side finalisation (worker2 of mem_process_seqs) -- filled in below the hot path; see DESIGN.md.
#include "bwahip_internal.h"
