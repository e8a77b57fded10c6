// development probe (never shipped): the 33-row contact kernel alone, device side only, for the compiler's resource report
//   hipcc --offload-arch=gfx950 --cuda-device-only -c -O3 ... -Rpass-analysis=kernel-resource-usage tools/probe_contact.hip -o /dev/null
#include <hip/hip_runtime.h>
#include "pdbatch.h"
#include "model.hpp"
#include <cmath>
#include <cstring>
#include "dev_const.hpp"
#include "reset_core.hpp"
#ifndef PDB_CONTACT_CPB
#define PDB_CONTACT_CPB 3
#endif
#ifndef PDB_KMINWAVES_C
#define PDB_KMINWAVES_C 2
#endif
#define PDB_KROWS 33
#define PDB_KMINWAVES 6
#define PDB_KSLOT0_EXACT true
#define PDB_KSLOT0_CTRL false
#define PDB_KCLASS_LS false
#define PDB_KERNEL_EXACT_C pdb_contact_kernel
#define PDB_KERNEL_GUARDED_C pdb_contact_kernel_generic
#define PDB_KERNEL_COLLIDE pdb_collide_kernel
#define PDB_KERNEL_EXACT_R pdb_resume_kernel
#define PDB_KERNEL_GUARDED_R pdb_resume_kernel_generic
#define PDB_KNS k33c
#define PDB_CPB PDB_CONTACT_CPB
#define PDB_HELPERS 0
#define PDB_SOLO 0
#define PDB_CONTACT_ONLY
#include "step_kernel.hip.inc"
