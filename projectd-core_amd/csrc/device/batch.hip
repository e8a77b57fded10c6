// pdbatch device batch: owns the HBM-resident car records of one GPU and launches the step kernel.
// See include/pdbatch.h for the reference interfaces each entry point replaces.
#include <hip/hip_runtime.h>
#include "pdbatch.h"
#include "model.hpp"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "dev_const.hpp"
#include "reset_core.hpp"   // Car::teleportByMode / Car::reset on a record: the host library's source, compiled for the device too

// two LDS size classes of the same kernel source: 33 constraint rows (the strut / live-axle and the all-double-wishbone
// cars; 8 KB of LDS per car, 6 workgroups per CU) and 40 rows (strut front + double wishbone rear: 38 rows; 5 per CU).
// Each class twice, so that the two passes can have workgroups of their own shape: PDB_FIRST_CPB / PDB_CONTACT_CPB cars per workgroup
// (car waves + the pack wave).  Measured in round 3 on the playground: the contact pass with one car per workgroup (no car waits for
// another at the workgroup's barriers) runs 393 / 628 us per launch against 188 / 333 us with three -- two-wave workgroups share
// their SIMDs with each other, and this pass is a few long single-wave chains.
#ifndef PDB_FIRST_CPB
#define PDB_FIRST_CPB 3
#endif
#ifndef PDB_CONTACT_CPB
#define PDB_CONTACT_CPB 3
#endif
// The contact pass's workgroup shape (round 5: four forms built, measured on the contact-heavy legs and on the headline, all bit-identical):
//   default            PDB_CONTACT_CPB = 3 car waves + a pack wave, every car's collision pass by its own wave (collisionGather + collisionNarrow<1>), side by side;
//   PDB_NARROW_COOP=1  the same workgroup, the narrow phase of one car at a time shared by all four waves;
//   PDB_CONTACT_CPB=1 PDB_CONTACT_HELPERS=2   one car per workgroup: its wave, a pack wave, two helper waves that work in the narrow phase only;
//   PDB_CONTACT_SOLO=1 one WAVE per car: the car's wave is its own pack wave, one role after the other (64-thread workgroups).
// In-kernel stamps say what each does to a car's chain (collision pass done 228 k / 86 k clocks after the car started for the coop forms, record stored 391 k /
// 140 k); the legs say it does not matter: playground MLP 35.2-36.3 / 33.4-33.7 / 32.9 M against 35.1-35.6 M for round 4's structure, the env loops and the
// headline within 2 % -- under the other partitions' first passes the pass is bound by what its 210-VGPR waves and 54 KB of LDS cost a CU while they live (a
// contact workgroup displaces two to three first-pass workgroups), not by the length of a car's chain.  The default is the form without
// any barrier of its own.
#ifndef PDB_CONTACT_SOLO
#define PDB_CONTACT_SOLO 0
#endif
#if PDB_CONTACT_SOLO
#undef PDB_CONTACT_CPB
#define PDB_CONTACT_CPB 1
#undef PDB_CONTACT_HELPERS
#define PDB_CONTACT_HELPERS 0
#define PDB_CONTACT_WAVES 1
#else
#ifndef PDB_CONTACT_HELPERS
#define PDB_CONTACT_HELPERS 0
#endif
#define PDB_CONTACT_WAVES (PDB_CONTACT_CPB + 1 + PDB_CONTACT_HELPERS)
#endif
#ifndef PDB_KMINWAVES_C
#define PDB_KMINWAVES_C 2
#endif
#ifndef PDB_COLLIDE_NW   /* waves of the collide kernel's one-car workgroup (its narrow phase is shared by all of them; the GPU is mostly empty while it runs) */
#define PDB_COLLIDE_NW PDB_CONTACT_WAVES
#endif
#define PDB_KROWS 33
#ifndef PDB_KMINWAVES33
#define PDB_KMINWAVES33 6
#endif
#define PDB_KMINWAVES PDB_KMINWAVES33
#define PDB_KERNEL_EXACT pdb_step_kernel
#define PDB_KSLOT0_EXACT true    /* the class's first kernel pair: exactly 33 rows, no run-time row guards */
#define PDB_KSLOT0_CTRL false
#define PDB_KCLASS_LS false     /* the per-lane setup table has a kernel pair of its own in this class (PDB_KERNEL_LANE) */
#define PDB_KERNEL_GUARDED pdb_step_kernel_generic
#define PDB_KERNEL_LANE pdb_step_kernel_lane_generic       /* the two forms reading the per-lane setup table (33-row class; the 40-row class tests for it at run time) */
#define PDB_KERNEL_LANE_EXACT pdb_step_kernel_lane
#define PDB_KERNEL_EXACT_C pdb_contact_kernel
#define PDB_KERNEL_GUARDED_C pdb_contact_kernel_generic
#define PDB_KERNEL_LANE_C pdb_contact_kernel_lane_generic
#define PDB_KERNEL_LANE_EXACT_C pdb_contact_kernel_lane
#define PDB_KERNEL_COLLIDE pdb_collide_kernel              /* the contact pass as two kernels (round 6): collision pass proper, one wave per car ... */
#define PDB_KERNEL_EXACT_R pdb_resume_kernel               /* ... and the tick's back half from the snapshot, in the first pass's workgroup shape */
#define PDB_KERNEL_GUARDED_R pdb_resume_kernel_generic
#define PDB_KERNEL_LANE_R pdb_resume_kernel_lane_generic
#define PDB_KERNEL_LANE_EXACT_R pdb_resume_kernel_lane
#define PDB_KNS k33
#define PDB_CPB PDB_FIRST_CPB
#define PDB_HELPERS 0
#define PDB_SOLO 0
#define PDB_FIRST_ONLY
#include "step_kernel.hip.inc"
#undef PDB_FIRST_ONLY
#undef PDB_SOLO
#undef PDB_HELPERS
#undef PDB_CPB
#undef PDB_BLOCK_THREADS
#undef PDB_KNS
#define PDB_KNS k33c
#define PDB_CPB PDB_CONTACT_CPB
#define PDB_HELPERS PDB_CONTACT_HELPERS
#define PDB_SOLO PDB_CONTACT_SOLO
#define PDB_CONTACT_ONLY
#include "step_kernel.hip.inc"
#undef PDB_CONTACT_ONLY
#undef PDB_SOLO
#undef PDB_HELPERS
#undef PDB_CPB
#undef PDB_BLOCK_THREADS
#undef PDB_KNS
#undef PDB_KROWS
#undef PDB_KMINWAVES
#undef PDB_KERNEL_EXACT
#undef PDB_KSLOT0_EXACT
#undef PDB_KSLOT0_CTRL
#undef PDB_KCLASS_LS
#undef PDB_KERNEL_GUARDED
#undef PDB_KERNEL_EXACT_C
#undef PDB_KERNEL_GUARDED_C
#undef PDB_KERNEL_LANE
#undef PDB_KERNEL_LANE_C
#undef PDB_KERNEL_LANE_EXACT
#undef PDB_KERNEL_LANE_EXACT_C
#undef PDB_KERNEL_COLLIDE
#undef PDB_KERNEL_EXACT_R
#undef PDB_KERNEL_GUARDED_R
#undef PDB_KERNEL_LANE_R
#undef PDB_KERNEL_LANE_EXACT_R
#ifndef PDB_FAST_BUILD   /* development builds (make dev) compile the 33-row size class only */
#define PDB_KROWS 40
#define PDB_KMINWAVES 5
#define PDB_KERNEL_EXACT pdb_step_kernel_ctrl
#define PDB_KSLOT0_EXACT false   /* the 40-row class's first kernel pair: row-guarded like the second, compiled with the DynamicController call sites (any car with controller files) */
#define PDB_KSLOT0_CTRL true
#define PDB_KCLASS_LS true      /* the 40-row class reads the per-lane setup table, where there is one, in its two kernel pairs (a run-time test) */
#define PDB_KERNEL_GUARDED pdb_step_kernel_wide
#define PDB_KERNEL_EXACT_C pdb_contact_kernel_ctrl
#define PDB_KERNEL_GUARDED_C pdb_contact_kernel_wide
#define PDB_KERNEL_COLLIDE pdb_collide_kernel_wide
#define PDB_KERNEL_EXACT_R pdb_resume_kernel_ctrl
#define PDB_KERNEL_GUARDED_R pdb_resume_kernel_wide
#define PDB_KNS k40
#define PDB_CPB PDB_FIRST_CPB
#define PDB_HELPERS 0
#define PDB_SOLO 0
#define PDB_FIRST_ONLY
#include "step_kernel.hip.inc"
#undef PDB_FIRST_ONLY
#undef PDB_SOLO
#undef PDB_HELPERS
#undef PDB_CPB
#undef PDB_BLOCK_THREADS
#undef PDB_KNS
#define PDB_KNS k40c
#define PDB_CPB PDB_CONTACT_CPB
#define PDB_HELPERS PDB_CONTACT_HELPERS
#define PDB_SOLO PDB_CONTACT_SOLO
#define PDB_CONTACT_ONLY
#include "step_kernel.hip.inc"
#undef PDB_CONTACT_ONLY
#undef PDB_SOLO
#undef PDB_HELPERS
#undef PDB_CPB
#undef PDB_BLOCK_THREADS
#undef PDB_KNS
#undef PDB_KROWS
#undef PDB_KMINWAVES
#undef PDB_KERNEL_EXACT
#undef PDB_KSLOT0_EXACT
#undef PDB_KSLOT0_CTRL
#undef PDB_KCLASS_LS
#undef PDB_KERNEL_GUARDED
#undef PDB_KERNEL_EXACT_C
#undef PDB_KERNEL_GUARDED_C
#undef PDB_KERNEL_COLLIDE
#undef PDB_KERNEL_EXACT_R
#undef PDB_KERNEL_GUARDED_R
/* exact-row class: the 26-row cars (no run-time row guards, the per-car LDS block sized for 26 rows) */
#define PDB_KROWS 26
#define PDB_KMINWAVES 6
#define PDB_KERNEL_EXACT pdb_step_kernel_r26
#define PDB_KSLOT0_EXACT true
#define PDB_KSLOT0_CTRL false
#define PDB_KCLASS_LS false
#define PDB_KERNEL_EXACT_C pdb_contact_kernel_r26
#define PDB_KERNEL_COLLIDE pdb_collide_kernel_r26
#define PDB_KERNEL_EXACT_R pdb_resume_kernel_r26
#define PDB_KNS k26
#define PDB_CPB PDB_FIRST_CPB
#define PDB_HELPERS 0
#define PDB_SOLO 0
#define PDB_FIRST_ONLY
#include "step_kernel.hip.inc"
#undef PDB_FIRST_ONLY
#undef PDB_SOLO
#undef PDB_HELPERS
#undef PDB_CPB
#undef PDB_BLOCK_THREADS
#undef PDB_KNS
#define PDB_KNS k26c
#define PDB_CPB PDB_CONTACT_CPB
#define PDB_HELPERS PDB_CONTACT_HELPERS
#define PDB_SOLO PDB_CONTACT_SOLO
#define PDB_CONTACT_ONLY
#include "step_kernel.hip.inc"
#undef PDB_CONTACT_ONLY
#undef PDB_SOLO
#undef PDB_HELPERS
#undef PDB_CPB
#undef PDB_BLOCK_THREADS
#undef PDB_KNS
#undef PDB_KROWS
#undef PDB_KMINWAVES
#undef PDB_KERNEL_EXACT
#undef PDB_KSLOT0_EXACT
#undef PDB_KSLOT0_CTRL
#undef PDB_KCLASS_LS
#undef PDB_KERNEL_EXACT_C
#undef PDB_KERNEL_COLLIDE
#undef PDB_KERNEL_EXACT_R
/* exact-row class: the 38-row cars (no run-time row guards, the per-car LDS block sized for 38 rows) */
#define PDB_KROWS 38
#define PDB_KMINWAVES 5
#define PDB_KERNEL_EXACT pdb_step_kernel_r38
#define PDB_KSLOT0_EXACT true
#define PDB_KSLOT0_CTRL false
#define PDB_KCLASS_LS true
#define PDB_KERNEL_EXACT_C pdb_contact_kernel_r38
#define PDB_KERNEL_COLLIDE pdb_collide_kernel_r38
#define PDB_KERNEL_EXACT_R pdb_resume_kernel_r38
#define PDB_KNS k38
#define PDB_CPB PDB_FIRST_CPB
#define PDB_HELPERS 0
#define PDB_SOLO 0
#define PDB_FIRST_ONLY
#include "step_kernel.hip.inc"
#undef PDB_FIRST_ONLY
#undef PDB_SOLO
#undef PDB_HELPERS
#undef PDB_CPB
#undef PDB_BLOCK_THREADS
#undef PDB_KNS
#define PDB_KNS k38c
#define PDB_CPB PDB_CONTACT_CPB
#define PDB_HELPERS PDB_CONTACT_HELPERS
#define PDB_SOLO PDB_CONTACT_SOLO
#define PDB_CONTACT_ONLY
#include "step_kernel.hip.inc"
#undef PDB_CONTACT_ONLY
#undef PDB_SOLO
#undef PDB_HELPERS
#undef PDB_CPB
#undef PDB_BLOCK_THREADS
#undef PDB_KNS
#undef PDB_KROWS
#undef PDB_KMINWAVES
#undef PDB_KERNEL_EXACT
#undef PDB_KSLOT0_EXACT
#undef PDB_KSLOT0_CTRL
#undef PDB_KCLASS_LS
#undef PDB_KERNEL_EXACT_C
#undef PDB_KERNEL_COLLIDE
#undef PDB_KERNEL_EXACT_R
#define PDB_EXACT_CLASSES 1
#endif
#undef PDB_KMINWAVES_C
#ifndef PDB_FAST_BUILD
static constexpr size_t kSnapStrideWide = k40::kSnapStride;
static_assert(k40::kSnapStride >= k33::kSnapStride && k33::kSnapStride == k33c::kSnapStride && k40::kSnapStride == k40c::kSnapStride, "snapshot slots");
static_assert(k40::kSnapStride >= k38::kSnapStride && k40::kSnapStride >= k26::kSnapStride && k26::kSnapStride == k26c::kSnapStride && k38::kSnapStride == k38c::kSnapStride, "snapshot slots of the exact-row classes");
#else
static constexpr size_t kSnapStrideWide = k33::kSnapStride;
#endif
#define PDB_CPB PDB_FIRST_CPB   /* host side: the first pass's workgroup */
#define PDB_BLOCK_THREADS (PDB_WAVE * (PDB_FIRST_CPB + 1))


namespace pdb { void setError(const std::string& s); }

#include <dlfcn.h>
// RCCL: types and prototypes only -- the library is taken at run time (rcclApi), never linked.  Where the headers are absent the few declarations the
// exchange uses are made here (the ABI of RCCL 2.x: a 128-byte id, an opaque communicator, int-sized enums), so the build does not depend on them either.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat = 7 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId*);
ncclResult_t ncclCommInitRank(ncclComm_t*, int, ncclUniqueId, int);
ncclResult_t ncclCommDestroy(ncclComm_t);
ncclResult_t ncclAllGather(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
ncclResult_t ncclSend(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
ncclResult_t ncclRecv(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
const char* ncclGetErrorString(ncclResult_t);
}
#endif
#define LAUNCHCHK(b) do { HIPCHK(hipGetLastError()); if ((b)->launchRefused) { (b)->launchRefused = false; pdb::setError("pdbatch: the tick was not launched: no hand-over snapshots for the contact pass (device memory; they are allocated by pdb_create / pdb_reset_mask_device / pdb_set_partition_params)"); return PDB_ERR_HIP; } } while (0)
#define HIPCHK(expr)                                                                                 \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            pdb::setError(std::string(#expr) + ": " + hipGetErrorString(_e));                        \
            return PDB_ERR_HIP;                                                                      \
        }                                                                                            \
    } while (0)

#define PDB_MAX_PARTS 4
#ifndef PDB_CONTACT_GRID
#define PDB_CONTACT_GRID 32
#endif
#ifndef PDB_CONTACT_GRID_IDLE
#define PDB_CONTACT_GRID_IDLE 1
#endif
#define PDB_KERNEL_SAMPLES 64
struct KernelSamples { hipEvent_t ev[2 * PDB_KERNEL_SAMPLES] = {}; int cars[PDB_KERNEL_SAMPLES] = {}; int n = 0; unsigned tick = 0; };
struct pdb_batch {
    int device = 0;
    int n = 0;
    pdb_car_params params;
    std::vector<uint8_t> track;
    DevConst K;
    hipStream_t stream = nullptr;
    pdb_dyn_state* dStates = nullptr;
    float* dActions = nullptr;
    int actionStride = 2;   // floats per car: 2 (CONTROLS / ENV) or 8 (FULL)
    pdb_step_out* dOut = nullptr;        // library-owned output block
    pdb_step_out* dOutActive = nullptr;  // where the next tick writes: dOut, or a caller-owned block (pdb_set_out_device)
    pdb_car_state* dCarStates = nullptr;
    pdb_car_params* dParams = nullptr;
    DevConst* dK = nullptr;
    uint8_t* dTrack = nullptr;
    // per-partition car blocks (pdb_set_partition_params: domain randomisation of tunes / scoring weights over the cars of one batch);
    // partHas[p] == false: the partition steps with the batch's block
    bool partHas[PDB_MAX_PARTS] = {false, false, false, false};
    pdb_car_params partParams[PDB_MAX_PARTS];
    DevConst partK[PDB_MAX_PARTS];
    pdb_car_params* dPartParams[PDB_MAX_PARTS] = {nullptr, nullptr, nullptr, nullptr};
    DevConst* dPartK[PDB_MAX_PARTS] = {nullptr, nullptr, nullptr, nullptr};
    uint8_t* dSnap = nullptr;   // [n][snapshot slot]: what the first pass leaves for the contact pass of the same tick per handed-over car (its LDS block at the force barrier)
    int* dQueue[PDB_MAX_PARTS + 1] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // RedoQueue per launch site: count, done, list[blocks]
    uint8_t* dResetScratch = nullptr;   // device copy of a host mask (pdb_reset)
    uint8_t* dResetMask = nullptr;   // [n]: 1 + teleport mode for cars to be reset at the top of their next tick (consumed and cleared by that tick)
    bool resetMaskArmed = false;     // pdb_reset_mask_device was asked for: the step kernels look at the mask
    unsigned char* dHold = nullptr;   // [n] hold mask of pdb_step_host_held, allocated on first use
    pdb_lane_setup* dLaneSetups = nullptr;   // [n] per-lane setup rows (pdb_set_lane_setups), allocated on first use and then complete: every lane's row holds its block's values or the caller's
    bool noExactClasses = false;
    float* dLawBias = nullptr;             // pdb_set_law's bias table where the library holds the copy (a caller's device table is used in place)
    pdb_lane_tune* dLaneTunes = nullptr;   // [n] per-lane tunes and reward weights (pdb_set_lane_tunes), allocated on first use; rows with valid == 0 fall back to the block
    pdb_slip_state* dSlip = nullptr;     // [2][n]: the cars' slipstreams (pdb_set_world_size; DevConst::slip)
    bool splitContact = true;            // the contact pass as the kernel pair where cars are expected in it (PDB_CONTACT_SPLIT=0: always the one kernel -- diagnostic A/B)
    pdb_dyn_state* dFresh = nullptr;     // device copy of resetTemplate (DevConst::freshState)
    pdb_dyn_state* dPartFresh[PDB_MAX_PARTS] = {nullptr, nullptr, nullptr, nullptr};   // a partition with a car block of its own: the fresh record of THAT block (ride height, pressures, fuel ...)
    pdb_contact* dContacts = nullptr;   // [n][PDB_MAX_CONTACTS]: each car's live contact joints (the first pdb_dyn_state.numContacts of its row)
    hipEvent_t ev0 = nullptr, ev1 = nullptr, tev0 = nullptr, tev1 = nullptr;
    bool ownStream = true;
    bool part0OnOwn = false;   // partition 0 runs on the batch's own stream (see pdb_set_partitions)
    double kernelMs = 0;
    int kernelLaunches = 0;
    // graph of `graphTicks` back-to-back ticks
    hipGraphExec_t graphExec = nullptr;
    int graphTicks = 0;
    float graphDt = 0;
    pdb_dyn_state resetTemplate;   // state of a fresh car teleported to the spline start
    unsigned long long* dStamps = nullptr;
    // free-running partitions (pdb_set_partitions / pdb_step_ring): contiguous car ranges, one stream each
    int parts = 1;
    hipStream_t partStream[PDB_MAX_PARTS] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t partFork = nullptr, partEnd[PDB_MAX_PARTS] = {nullptr, nullptr, nullptr, nullptr}, partStart[PDB_MAX_PARTS] = {nullptr, nullptr, nullptr, nullptr};
    bool partMark = false;
    bool partEndStale[PDB_MAX_PARTS] = {false, false, false, false};   // work enqueued on the partition's stream since its end event was last recorded (pdb_step_partition records it
                                                                      // only when somebody asks: one runtime call less per partition and tick in the launch-bound closed loops)
    bool partDirty = false;   // partition kernels enqueued that the batch's stream has not been ordered after
    bool batchDirty = false;  // asynchronous work queued on the batch's stream that pdb_step_partition's streams have not been ordered after
    float* hActions = nullptr;          // page-locked host mirrors (pdb_host_actions / pdb_host_out): the pipelined host-policy loop
    pdb_step_out* hOut = nullptr;
    int contactGrid = 0;   // workgroups of the contact pass: 0 = adaptive (from the queue lengths the last passes saw); PDB_CONTACT_GRID in the environment fixes it (diagnostic)
    int burst[PDB_MAX_PARTS + 1] = {0, 0, 0, 0, 0};   // per launch site: ticks for which the contact pass is launched wide whatever the hint says (after a reset: cars just put down
                                                        // tend to go through the pass once, all of them in the same tick)
    bool launchRefused = false;   // a tick could not be launched (no memory for the hand-over snapshots): the entry point that asked returns PDB_ERR_HIP
    bool capturing = false;   // launches being recorded into a graph: the contact pass's grid is then frozen, so it is not sized for an idle pass
    // the per-tick exchange with the learner issued by the library itself (pdb_comm_init / pdb_step_exchange_partition): one RCCL communicator per partition
    ncclComm_t comm[PDB_MAX_PARTS] = {nullptr, nullptr, nullptr, nullptr};
    int commWorld = 0, commRank = 0;
    int sampleEvery = 0;   // pdb_sample_kernel: events around every k-th first-pass launch of each launch site
    KernelSamples samples[PDB_MAX_PARTS + 1];
    int* hHint = nullptr;  // page-locked, device-visible: per launch site, the number of cars the last contact pass held (written by its last workgroup; read here without waiting)
    int* dHint = nullptr;
};
static int partFirst(const pdb_batch* b, int p);
static void commFree(pdb_batch* b);
static void wideContactPasses(pdb_batch* b);
// Every entry point that works through the batch's stream first lets that stream wait for the partitions' kernels still in
// flight (pdb_step_ring with join == 0): state reads, resets and plain launches are always ordered after them.
static int freshEnd(pdb_batch* b, int p) {   // the partition's end event, brought up to what has been enqueued on its stream
    if (b->partEndStale[p] && b->partEnd[p] && b->partStream[p]) { HIPCHK(hipEventRecord(b->partEnd[p], b->partStream[p])); b->partEndStale[p] = false; }
    return PDB_OK;
}
static int joinParts(pdb_batch* b) {
    if (b->parts > 1 && b->partDirty) {
        for (int p = 0; p < b->parts; ++p)
            if (b->partEnd[p] && partFirst(b, p + 1) > partFirst(b, p)) { if (int rcf = freshEnd(b, p)) return rcf; HIPCHK(hipStreamWaitEvent(b->stream, b->partEnd[p], 0)); }
        b->partDirty = false;
    }
    return PDB_OK;
}
static int partFirst(const pdb_batch* b, int p) {   // boundaries on whole workgroups
    if (p >= b->parts) return b->n;
    const long long raw = (long long)b->n * p / b->parts;
    const int unit = PDB_CPB * (b->K.worldSize > 1 ? b->K.worldSize : 1);   // ... and, in a batch of multi-car simulators, on whole worlds
    return (int)(raw / unit * unit);
}

static void fillConst(const pdb_car_params& P, DevConst& K, int actionMode) {
    memset(&K, 0, sizeof(K));
    for (int i = 0; i < 4; ++i) {
        // mat44f::createFromAxisAngle((0,0,1), staticCamber): M11 = M22 = c, M12 = s, M21 = -s, M33 = (1-c)+c
        const float s = pm::sinf_(P.susp[i].staticCamber), c = pm::cosf_(P.susp[i].staticCamber), o = 1.0f - c;
        K.camC[i] = ((0.0f * 0.0f) * o) + c;
        K.camS[i] = (1.0f * s) + (0.0f * 0.0f) * o;
        K.camM33[i] = ((1.0f * 1.0f) * o) + c;
    }
    K.acos096 = pm::acosf_(0.96f);
    int r = 0;
    for (int j = 0; j < P.numJoints; ++j) {
        K.rowStart[j] = r;
        const int t = P.joints[j].type;
        r += (t == PDB_JOINT_FIXED) ? 6 : (t == PDB_JOINT_BALL) ? 3 : (t == PDB_JOINT_SLIDER) ? 5 : 1;
    }
    for (int j = P.numJoints; j <= PDB_MAX_JOINTS; ++j) K.rowStart[j] = r;
    for (int j = 0; j < P.numJoints; ++j)
        for (int rr = K.rowStart[j]; rr < K.rowStart[j + 1] && rr < PDB_MAX_ROWS; ++rr) { K.rowB0[rr] = P.joints[j].b0; K.rowB1[rr] = P.joints[j].b1; K.rowFirst[rr] = (rr == K.rowStart[j]) ? 1 : 0; }
    static_assert(PDB_MAX_ROWS <= 48, "cfOff row");
    for (int b = 0; b < PDB_MAX_BODIES; ++b)
        for (int rr = 0; rr < 48; ++rr)
            K.cfOff[b][rr] = (rr < r && rr < PDB_MAX_ROWS) ? (K.rowB0[rr] == b ? 0 : (K.rowB1[rr] == b ? 24 : 255)) : 255;
    for (int b = 0; b < PDB_MAX_BODIES; ++b) {
        K.invMass[b] = (b < P.numBodies) ? 1.0f / P.bodies[b].mass : 0.0f;
        for (int k = 0; k < 3; ++k) K.invInertia[b][k] = (b < P.numBodies) ? 1.0f / P.bodies[b].inertia[k] : 0.0f;
    }
    (void)pdb_lane_tune_from_params(&P, &K.laneDefault);
    K.laneTunes = nullptr;
    K.laneSetups = nullptr;
    K.actionMode = actionMode;
    K.wantCarState = 0;
    K.stuckTimeout = 5.0;   // projectd_env.py:47
}

// One tick of the cars [c0, c1) on `st`: the first pass over every car, then -- when the car model has body colliders -- the
// contact pass over the blocks the first pass queued (cars with live contact joints or fresh contacts; a small fixed grid that
// finds an empty queue on almost every tick).  `q` = which of the batch's queues this launch site uses (one per partition
// stream, one for the batch's own stream: launches that can be in flight together never share a queue).
// The hand-over snapshots (about 10 KB per car, four times the record) exist only once a contact pass can run: body colliders on, the
// reset mask armed or the in-tick auto-teleport set.  Allocated on first need, outside any graph capture (pdb_step_n calls this before it records).
static bool passNeeded(const pdb_batch* b, const pdb_car_params& HP) { return HP.collider.enabled != 0 || b->resetMaskArmed || HP.autoTeleport != 0; }
// Called by the entry points that can turn a contact pass on (pdb_create, pdb_reset_mask_device, pdb_set_partition_params), never from a launch: a
// hipMalloc inside a caller's stream capture would invalidate the capture, and running out of memory belongs to set-up, not to the middle of a run.
static void ensureSnap(pdb_batch* b) {
    if (b->dSnap) return;
    bool need = passNeeded(b, b->params);
    for (int p = 0; p < PDB_MAX_PARTS; ++p) need = need || (b->partHas[p] && passNeeded(b, b->partParams[p]));
    if (need && hipMalloc(&b->dSnap, kSnapStrideWide * (size_t)b->n) != hipSuccess) { b->dSnap = nullptr; pdb::setError("pdbatch: out of device memory for the contact pass's hand-over snapshots"); }
}
static void launchTick(pdb_batch* b, hipStream_t st, int c0, int c1, pdb_step_out* out, int q) {
    const int nblk = (c1 - c0 + PDB_CPB - 1) / PDB_CPB, m = b->params.numRows;
    pdb_dyn_state* S = b->dStates + c0;
    const float* Aact = b->dActions + (size_t)c0 * b->actionStride;
    pdb_step_out* O = out + c0;
    pdb_car_state* CS = b->dCarStates ? b->dCarStates + c0 : nullptr;
    pdb_contact* CT = b->dContacts + (size_t)c0 * PDB_MAX_CONTACTS;
    void* Q = b->dQueue[q];
    const bool own = q < PDB_MAX_PARTS && b->partHas[q];
    const pdb_car_params* DP = own ? b->dPartParams[q] : b->dParams;
    const DevConst* DK = (q < PDB_MAX_PARTS && b->dPartK[q]) ? b->dPartK[q] : b->dK;   // a partition launch: the block whose per-car tables start at the partition's first car
    const pdb_car_params& HP = own ? b->partParams[q] : b->params;
    // every launch's snapshot region starts at the ALLOCATION's stride (partitions may carry car blocks of different kernel classes: with the
    // launch's own, narrower stride a later partition's region would begin inside an earlier, wider one's); the kernel indexes its slots with its own stride inside it
    uint8_t* SN = b->dSnap + (size_t)c0 * kSnapStrideWide;
    uint8_t* RM = b->resetMaskArmed ? b->dResetMask + c0 : nullptr;
    const int n = c1 - c0;
    // the contact pass also serves episode resets asked for through the reset mask and the in-tick auto-teleport
    const bool contacts = passNeeded(b, HP);
    if (contacts && !b->dSnap) { b->launchRefused = true; return; }   // never a first pass whose queued cars nobody finishes (LAUNCHCHK reports it once and clears the flag)
    // the contact pass's grid: enough workgroups for twice the cars its last pass held (a stale number, read without waiting: it only
    // sizes the grid -- workgroups take the queued cars in turn whatever their number), at least PDB_CONTACT_GRID, at most what is resident at once
    int cg = b->contactGrid;
    if (cg <= 0) {
        const int held = b->hHint ? *(volatile int*)(b->hHint + q) : 0;
        cg = (2 * held + PDB_CONTACT_CPB - 1) / PDB_CONTACT_CPB;
        if (cg < (held > 0 ? PDB_CONTACT_GRID : PDB_CONTACT_GRID_IDLE)) cg = held > 0 ? PDB_CONTACT_GRID : PDB_CONTACT_GRID_IDLE;   // nobody touched anything lately: a handful of workgroups is launched, found empty and gone
        if (b->capturing && cg < PDB_CONTACT_GRID) cg = PDB_CONTACT_GRID;   // a replayed graph cannot follow the load
        // right after a reset of many cars: room for 1536 queued cars whatever the hint says (PDB_CONTACT_CPB cars per workgroup), never more than 4096 cars' worth
        if (!b->capturing && b->burst[q] > 0) { --b->burst[q]; if (cg < (1536 + PDB_CONTACT_CPB - 1) / PDB_CONTACT_CPB) cg = (1536 + PDB_CONTACT_CPB - 1) / PDB_CONTACT_CPB; }
        if (cg > (4096 + PDB_CONTACT_CPB - 1) / PDB_CONTACT_CPB) cg = (4096 + PDB_CONTACT_CPB - 1) / PDB_CONTACT_CPB;
    }
    int* HN = b->dHint ? b->dHint + q : nullptr;
    const dim3 grid(nblk), block(PDB_BLOCK_THREADS), cblock(PDB_WAVE * PDB_CONTACT_WAVES), cgrid(((n + PDB_CONTACT_CPB - 1) / PDB_CONTACT_CPB) < cg ? ((n + PDB_CONTACT_CPB - 1) / PDB_CONTACT_CPB) : cg);
    // a car with DynamicController files goes through the kernel pair compiled with the controllers' call sites (40-row class, row-guarded: any car)
    const bool ctrl = HP.numCtrlStages != 0 || HP.hasBrakeTemps != 0;
    // the exact-row kernels hold the team form of the car waves' stage only (step_kernel.hip.inc): a model with more joints than a car's share of the wave's lanes, or a batch
    // created under PDB_NO_TEAM, steps through the row-guarded kernels
    const bool teamOk = HP.numJoints <= PDB_WAVE / PDB_FIRST_CPB && b->K.noTeam == 0;
#ifdef PDB_EXACT_CLASSES   /* 6 / 7: the classes compiled for exactly 26 / 38 rows (the other shipped cars: double wishbones all round with and without a multilink-style rear) */
    const int kind = ctrl ? 0 : (m > 33 ? ((m == 38 && !b->noExactClasses && teamOk) ? 7 : 3) : (b->dLaneSetups ? ((m == 33 && teamOk) ? 5 : 4) : ((m == 33 && teamOk) ? 1 : ((m == 26 && !b->noExactClasses && teamOk) ? 6 : 2))));
#else
    const int kind = ctrl ? 0 : (m > 33 ? 3 : (b->dLaneSetups ? ((m == 33 && teamOk) ? 5 : 4) : ((m == 33 && teamOk) ? 1 : 2)));
#endif   // (4: the 33-row class's kernel pair compiled for the per-lane setup table; the 40-row class tests for the table at run time)
    // measurement (pdb_sample_kernel): HIP events around every k-th first-pass launch of this site, on the stream it is launched on
    KernelSamples& KS = b->samples[q];
    const bool sampled = b->sampleEvery > 0 && !b->capturing && KS.ev[0] && (KS.tick++ % b->sampleEvery) == 0 && KS.n < PDB_KERNEL_SAMPLES;
    if (sampled) (void)hipEventRecord(KS.ev[2 * KS.n], st);
    switch (kind) {
#ifndef PDB_FAST_BUILD
    case 0: hipLaunchKernelGGL(k40::pdb_step_kernel_ctrl, grid, block, 0, st, S, Aact, O, CS, DP, DK, b->dTrack, (k40::RedoQueue*)Q, RM, n, SN); break;
    case 6: hipLaunchKernelGGL(k26::pdb_step_kernel_r26, grid, block, 0, st, S, Aact, O, CS, DP, DK, b->dTrack, (k26::RedoQueue*)Q, RM, n, SN); break;
    case 7: hipLaunchKernelGGL(k38::pdb_step_kernel_r38, grid, block, 0, st, S, Aact, O, CS, DP, DK, b->dTrack, (k38::RedoQueue*)Q, RM, n, SN); break;
    case 3: hipLaunchKernelGGL(k40::pdb_step_kernel_wide, grid, block, 0, st, S, Aact, O, CS, DP, DK, b->dTrack, (k40::RedoQueue*)Q, RM, n, SN); break;
#endif
    case 1: hipLaunchKernelGGL(k33::pdb_step_kernel, grid, block, 0, st, S, Aact, O, CS, DP, DK, b->dTrack, (k33::RedoQueue*)Q, RM, n, SN); break;
    case 2: hipLaunchKernelGGL(k33::pdb_step_kernel_generic, grid, block, 0, st, S, Aact, O, CS, DP, DK, b->dTrack, (k33::RedoQueue*)Q, RM, n, SN); break;
    case 4: hipLaunchKernelGGL(k33::pdb_step_kernel_lane_generic, grid, block, 0, st, S, Aact, O, CS, DP, DK, b->dTrack, (k33::RedoQueue*)Q, RM, n, SN); break;
    case 5: hipLaunchKernelGGL(k33::pdb_step_kernel_lane, grid, block, 0, st, S, Aact, O, CS, DP, DK, b->dTrack, (k33::RedoQueue*)Q, RM, n, SN); break;
    default: break;
    }
    if (sampled) { (void)hipEventRecord(KS.ev[2 * KS.n + 1], st); KS.cars[KS.n] = n; ++KS.n; }
    if (!contacts) return;
    // The contact pass, in one of two forms with the same results (step_kernel.hip.inc, pdbCollideBody): where cars are expected in it -- the hint of the last passes, or a
    // caller's fixed grid -- the snapshot cars go through the kernel pair sized to fit beside the first pass (collide: one wave per car; resume: the first pass's workgroup
    // shape); the one kernel follows only if a car's whole tick may have to be done again (reset mask, in-tick auto-teleport: its other list).  Where nothing has been
    // touched lately the one kernel alone: its idle launch ends on three words, and a surprise is still served (and raises the hint).
    // (A launch recorded into a graph keeps the form chosen at capture: a caller that captures pdb_step_partition asks for a fixed grid, pdb_set_contact_grid, and so for the pair.)
    const int heldNow = (b->contactGrid > 0) ? 1 : (b->hHint ? *(volatile int*)(b->hHint + q) : 0);
    if (b->splitContact && HP.collider.enabled != 0 && heldNow > 0) {
        const dim3 xgrid((unsigned)(n < (int)cgrid.x * PDB_CONTACT_CPB ? n : (int)cgrid.x * PDB_CONTACT_CPB))   /* one car per workgroup */, xblock(PDB_WAVE * PDB_COLLIDE_NW);
        switch (kind) {
#ifndef PDB_FAST_BUILD
        case 0: case 3: hipLaunchKernelGGL(k40c::pdb_collide_kernel_wide, xgrid, xblock, 0, st, DP, b->dTrack, CT, (k40c::RedoQueue*)Q, SN); break;
        case 6: hipLaunchKernelGGL(k26c::pdb_collide_kernel_r26, xgrid, xblock, 0, st, DP, b->dTrack, CT, (k26c::RedoQueue*)Q, SN); break;
        case 7: hipLaunchKernelGGL(k38c::pdb_collide_kernel_r38, xgrid, xblock, 0, st, DP, b->dTrack, CT, (k38c::RedoQueue*)Q, SN); break;
#endif
        default: hipLaunchKernelGGL(k33c::pdb_collide_kernel, xgrid, xblock, 0, st, DP, b->dTrack, CT, (k33c::RedoQueue*)Q, SN); break;
        }
        switch (kind) {
#ifndef PDB_FAST_BUILD
        case 0: hipLaunchKernelGGL(k40c::pdb_resume_kernel_ctrl, cgrid, cblock, 0, st, S, Aact, O, CS, DP, DK, b->dTrack, CT, (k40c::RedoQueue*)Q, n, HN, SN); break;
        case 3: hipLaunchKernelGGL(k40c::pdb_resume_kernel_wide, cgrid, cblock, 0, st, S, Aact, O, CS, DP, DK, b->dTrack, CT, (k40c::RedoQueue*)Q, n, HN, SN); break;
        case 6: hipLaunchKernelGGL(k26c::pdb_resume_kernel_r26, cgrid, cblock, 0, st, S, Aact, O, CS, DP, DK, b->dTrack, CT, (k26c::RedoQueue*)Q, n, HN, SN); break;
        case 7: hipLaunchKernelGGL(k38c::pdb_resume_kernel_r38, cgrid, cblock, 0, st, S, Aact, O, CS, DP, DK, b->dTrack, CT, (k38c::RedoQueue*)Q, n, HN, SN); break;
#endif
        case 1: hipLaunchKernelGGL(k33c::pdb_resume_kernel, cgrid, cblock, 0, st, S, Aact, O, CS, DP, DK, b->dTrack, CT, (k33c::RedoQueue*)Q, n, HN, SN); break;
        case 2: hipLaunchKernelGGL(k33c::pdb_resume_kernel_generic, cgrid, cblock, 0, st, S, Aact, O, CS, DP, DK, b->dTrack, CT, (k33c::RedoQueue*)Q, n, HN, SN); break;
        case 4: hipLaunchKernelGGL(k33c::pdb_resume_kernel_lane_generic, cgrid, cblock, 0, st, S, Aact, O, CS, DP, DK, b->dTrack, CT, (k33c::RedoQueue*)Q, n, HN, SN); break;
        case 5: hipLaunchKernelGGL(k33c::pdb_resume_kernel_lane, cgrid, cblock, 0, st, S, Aact, O, CS, DP, DK, b->dTrack, CT, (k33c::RedoQueue*)Q, n, HN, SN); break;
        default: break;
        }
        if (!(b->resetMaskArmed || HP.autoTeleport != 0)) return;   // no car's whole tick is ever done again: the queue's other list stays empty
    }
    switch (kind) {
#ifndef PDB_FAST_BUILD
    case 0: hipLaunchKernelGGL(k40c::pdb_contact_kernel_ctrl, cgrid, cblock, 0, st, S, Aact, O, CS, DP, DK, b->dTrack, CT, (k40c::RedoQueue*)Q, RM, n, HN, SN); break;
    case 3: hipLaunchKernelGGL(k40c::pdb_contact_kernel_wide, cgrid, cblock, 0, st, S, Aact, O, CS, DP, DK, b->dTrack, CT, (k40c::RedoQueue*)Q, RM, n, HN, SN); break;
    case 6: hipLaunchKernelGGL(k26c::pdb_contact_kernel_r26, cgrid, cblock, 0, st, S, Aact, O, CS, DP, DK, b->dTrack, CT, (k26c::RedoQueue*)Q, RM, n, HN, SN); break;
    case 7: hipLaunchKernelGGL(k38c::pdb_contact_kernel_r38, cgrid, cblock, 0, st, S, Aact, O, CS, DP, DK, b->dTrack, CT, (k38c::RedoQueue*)Q, RM, n, HN, SN); break;
#endif
    case 1: hipLaunchKernelGGL(k33c::pdb_contact_kernel, cgrid, cblock, 0, st, S, Aact, O, CS, DP, DK, b->dTrack, CT, (k33c::RedoQueue*)Q, RM, n, HN, SN); break;
    case 2: hipLaunchKernelGGL(k33c::pdb_contact_kernel_generic, cgrid, cblock, 0, st, S, Aact, O, CS, DP, DK, b->dTrack, CT, (k33c::RedoQueue*)Q, RM, n, HN, SN); break;
    case 4: hipLaunchKernelGGL(k33c::pdb_contact_kernel_lane_generic, cgrid, cblock, 0, st, S, Aact, O, CS, DP, DK, b->dTrack, CT, (k33c::RedoQueue*)Q, RM, n, HN, SN); break;
    case 5: hipLaunchKernelGGL(k33c::pdb_contact_kernel_lane, cgrid, cblock, 0, st, S, Aact, O, CS, DP, DK, b->dTrack, CT, (k33c::RedoQueue*)Q, RM, n, HN, SN); break;
    default: break;
    }
}

// One tick of every car on `st` (the whole-batch entry points: pdb_step, pdb_step_n and its graph, pdb_step_async, pdb_step_host).
// Partitions with a car block of their own (pdb_set_partition_params) step with it here too: one launch per partition range.
static void launchWhole(pdb_batch* b, hipStream_t st, pdb_step_out* out) {
    bool any = false;
    for (int p = 0; p < b->parts; ++p) any = any || b->partHas[p];
    if (!any) { launchTick(b, st, 0, b->n, out, PDB_MAX_PARTS); return; }
    for (int p = 0; p < b->parts; ++p) {
        const int c0 = partFirst(b, p), c1 = partFirst(b, p + 1);
        if (c1 > c0) launchTick(b, st, c0, c1, out, p);   // the partition's own queue: its stream is joined (joinParts) before anything is launched here
    }
}

// the constants blocks to the device: the batch's, and every partition's own (its model-derived part + the batch's run-time part)
static int pushK(pdb_batch* b, hipStream_t st, bool async) {
    if (async) HIPCHK(hipMemcpyAsync(b->dK, &b->K, sizeof(DevConst), hipMemcpyHostToDevice, st)); else HIPCHK(hipMemcpy(b->dK, &b->K, sizeof(DevConst), hipMemcpyHostToDevice));
    // every partition has a constants block of its own on the device: its car block's model-derived part where it has one (else the batch's), and the
    // batch's per-car tables (lane tunes, hold mask) offset to the partition's first car -- a launch over the cars [c0, c1) indexes everything from c0
    for (int p = 0; p < b->parts && p < PDB_MAX_PARTS; ++p) {
        if (!b->dPartK[p]) continue;   // (one part: no partition launches)
        DevConst& K = b->partK[p];
        if (b->partHas[p]) fillConst(b->partParams[p], K, b->K.actionMode); else K = b->K;
        const int c0 = partFirst(b, p);
        K.freshState = (b->partHas[p] && b->dPartFresh[p]) ? b->dPartFresh[p] : b->K.freshState; K.noTeam = b->K.noTeam;
        K.laneTunes = b->K.laneTunes ? b->K.laneTunes + c0 : nullptr;
        K.laneSetups = b->K.laneSetups ? b->K.laneSetups + c0 : nullptr;
        K.holdMask = b->K.holdMask ? b->K.holdMask + c0 : nullptr;
        K.worldSize = b->K.worldSize; K.slipStride = b->K.slipStride; K.slip = b->K.slip ? b->K.slip + c0 : nullptr;
        K.lawPeriod = b->K.lawPeriod; K.lawStride = b->K.lawStride; K.lawBias = b->K.lawBias ? b->K.lawBias + 2 * (size_t)c0 : nullptr;
        memcpy(K.lawBias0, b->K.lawBias0, sizeof(K.lawBias0)); memcpy(K.lawW, b->K.lawW, sizeof(K.lawW));
        K.dt = b->K.dt; K.fps = b->K.fps; K.dtD = b->K.dtD; K.stuckTimeout = b->K.stuckTimeout; K.wantCarState = b->K.wantCarState; K.stamps = b->K.stamps; K.stampCars = b->K.stampCars;
        K.envHitPenalty = b->K.envHitPenalty; K.envOffPenalty = b->K.envOffPenalty; K.envStuckPenalty = b->K.envStuckPenalty; K.envLowReward = b->K.envLowReward;
        K.envMode = b->K.envMode; K.envTermHit = b->K.envTermHit; K.envTermOff = b->K.envTermOff; K.envTermStuck = b->K.envTermStuck;
        K.envTeleportOnReset = b->K.envTeleportOnReset; K.envTeleportMode = b->K.envTeleportMode;
        if (async) HIPCHK(hipMemcpyAsync(b->dPartK[p], &K, sizeof(DevConst), hipMemcpyHostToDevice, st)); else HIPCHK(hipMemcpy(b->dPartK[p], &K, sizeof(DevConst), hipMemcpyHostToDevice));
    }
    return PDB_OK;
}

static int launch(pdb_batch* b, float dt, bool wantCarState) {
    if (int rcj = joinParts(b)) return rcj;
    if (b->K.dt != dt || b->K.wantCarState != (wantCarState ? 1 : 0)) {
        b->K.dt = dt; b->K.fps = 1.0f / dt;
        b->K.dtD = (dt == (float)(1.0 / 333.0)) ? (1.0 / 333.0) : (double)dt;   // PyProjectD.cpp:160-173: double dt, float step
        b->K.wantCarState = wantCarState ? 1 : 0;
        if (int rck = pushK(b, b->stream, true)) return rck;
    }
    launchWhole(b, b->stream, b->dOutActive);
    LAUNCHCHK(b);
    return PDB_OK;
}

extern "C" {

pdb_batch* pdb_create(int device, int n_cars, const pdb_car_params* params, const void* track_blob, uint64_t track_bytes, int action_mode) {
    if (!params || !track_blob || n_cars <= 0) { pdb::setError("pdb_create: bad argument"); return nullptr; }
    if (action_mode < PDB_ACTION_CONTROLS || action_mode > PDB_ACTION_FULL) { pdb::setError("pdb_create: unknown action mode"); return nullptr; }
    if (params->magic != 0x50434450 || params->version != 1 || params->numBodies < 2 || params->numBodies > PDB_MAX_BODIES ||
        params->numJoints < 1 || params->numJoints > PDB_MAX_JOINTS || params->numRows < 1 || params->numRows > PDB_MAX_ROWS ||
        params->numWings < 0 || params->numWings > PDB_MAX_WINGS || params->numTurbos < 0 || params->numTurbos > PDB_MAX_TURBOS) {
        pdb::setError("pdb_create: not a pdb_car_params block of this version (build it with pdb_build_car_model)"); return nullptr;
    }
    {
        const pdb_track_header* th = static_cast<const pdb_track_header*>(track_blob);
        if (track_bytes < sizeof(pdb_track_header) || th->magic != 0x4B544450 || th->version != 6 || th->totalBytes != track_bytes) {
            pdb::setError("pdb_create: not a track blob of this version (build it with pdb_build_track)"); return nullptr;
        }
    }
#ifdef PDB_FAST_BUILD   // the development build carries the 33-row class only: refuse what it cannot step rather than return PDB_OK without a launch
    if (params->numRows > 33 || params->numCtrlStages != 0 || params->hasBrakeTemps != 0) { pdb::setError("pdb_create: this development build (PDB_FAST_BUILD) holds only the 33-row kernels without controllers"); return nullptr; }
#endif
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) {
        pdb::setError("pdb_create: no usable HIP device (there is no CPU fallback)");
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) { pdb::setError("hipSetDevice failed"); return nullptr; }
    pdb_batch* b = new pdb_batch();
    b->device = device; b->n = n_cars; b->params = *params;
    if (const char* cg = getenv("PDB_CONTACT_GRID")) { const int v = atoi(cg); if (v > 0) b->contactGrid = v; }
    b->track.assign((const uint8_t*)track_blob, (const uint8_t*)track_blob + track_bytes);
    fillConst(b->params, b->K, action_mode);
    b->K.dt = (float)(1.0 / 333.0); b->K.fps = 1.0f / b->K.dt; b->K.dtD = 1.0 / 333.0;
    {   // which form of the contact pass where cars are expected in it (launchTick).  The kernel pair pays where a car's collision pass is long -- its narrow phase
        // is shared by four waves there -- and costs a launch and a trip of the snapshot through HBM where it is short: measured (profiles/r06_contact_split.txt) +3.5 % on
        // the playground-scale meshes (obstacles of centimetre-sized triangles: grid cells with hundreds of entries), -4 % on a wall-lined road of five-metre triangles.
        // The static proxy: the fullest cell of the collision grid.
        const pdb_track_header* th = reinterpret_cast<const pdb_track_header*>(b->track.data());
        int fullest = 0;
        if (th->gridNx > 0 && th->gridNz > 0) {
            const int32_t* gs = reinterpret_cast<const int32_t*>(b->track.data() + th->offGridStart);
            const int64_t cells = (int64_t)th->gridNx * th->gridNz;
            for (int64_t c = 0; c < cells; ++c) { const int e = gs[c + 1] - gs[c]; if (e > fullest) fullest = e; }
        }
        b->splitContact = fullest >= 48;
    }
    if (const char* sp = getenv("PDB_CONTACT_SPLIT")) b->splitContact = atoi(sp) != 0;   // diagnostic: 0 = the one-kernel contact pass always, 1 = the pair wherever cars are expected (tests, A/B)
    if (const char* ne = getenv("PDB_NO_EXACT_CLASSES")) b->noExactClasses = atoi(ne) != 0;   // diagnostic: the 26- / 38-row cars through the row-guarded kernels of the 33- / 40-row classes, as before round 6 (tests, A/B)
    if (const char* nt = getenv("PDB_NO_TEAM")) b->K.noTeam = atoi(nt) != 0 ? 1 : 0;   // diagnostic: the per-wave form of the car waves' stage (tests, A/B)
    bool ok = true;
    ok = ok && hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipMalloc(&b->dStates, sizeof(pdb_dyn_state) * (size_t)n_cars) == hipSuccess;
    b->actionStride = (action_mode == PDB_ACTION_FULL) ? 8 : 2;
    ok = ok && hipMalloc(&b->dActions, sizeof(float) * b->actionStride * (size_t)n_cars) == hipSuccess;
    ok = ok && hipMalloc(&b->dOut, sizeof(pdb_step_out) * (size_t)n_cars) == hipSuccess;
    ok = ok && hipMalloc(&b->dCarStates, sizeof(pdb_car_state) * (size_t)n_cars) == hipSuccess;
    ok = ok && hipMalloc(&b->dParams, sizeof(pdb_car_params)) == hipSuccess;
    ok = ok && hipMalloc(&b->dK, sizeof(DevConst)) == hipSuccess;
    ok = ok && hipMalloc(&b->dTrack, track_bytes) == hipSuccess;
    ok = ok && hipMalloc(&b->dContacts, sizeof(pdb_contact) * PDB_MAX_CONTACTS * (size_t)n_cars) == hipSuccess;
    ok = ok && hipMemset(b->dContacts, 0, sizeof(pdb_contact) * PDB_MAX_CONTACTS * (size_t)n_cars) == hipSuccess;
    ok = ok && hipMalloc(&b->dResetMask, (size_t)n_cars) == hipSuccess;
    ok = ok && hipMemset(b->dResetMask, 0, (size_t)n_cars) == hipSuccess;
    for (int q = 0; q <= PDB_MAX_PARTS; ++q) {
        const size_t qb = sizeof(int) * (size_t)(4 + n_cars);   // count, done, countFull, pad, one entry per car
        ok = ok && hipMalloc(&b->dQueue[q], qb) == hipSuccess;
        ok = ok && hipMemset(b->dQueue[q], 0, qb) == hipSuccess;
    }
    ok = ok && hipHostMalloc((void**)&b->hHint, sizeof(int) * (PDB_MAX_PARTS + 1), hipHostMallocMapped) == hipSuccess;
    if (ok) { memset(b->hHint, 0, sizeof(int) * (PDB_MAX_PARTS + 1)); ok = hipHostGetDevicePointer((void**)&b->dHint, b->hHint, 0) == hipSuccess; }
    ok = ok && hipEventCreate(&b->ev0) == hipSuccess && hipEventCreate(&b->ev1) == hipSuccess;
    ok = ok && hipEventCreate(&b->tev0) == hipSuccess && hipEventCreate(&b->tev1) == hipSuccess;
    if (ok) {
        ok = ok && hipMemcpy(b->dParams, &b->params, sizeof(pdb_car_params), hipMemcpyHostToDevice) == hipSuccess;
        ok = ok && hipMemcpy(b->dK, &b->K, sizeof(DevConst), hipMemcpyHostToDevice) == hipSuccess;
        ok = ok && hipMemcpy(b->dTrack, b->track.data(), track_bytes, hipMemcpyHostToDevice) == hipSuccess;
        ok = ok && hipMemset(b->dActions, 0, sizeof(float) * b->actionStride * (size_t)n_cars) == hipSuccess;
        ok = ok && hipMemset(b->dOut, 0, sizeof(pdb_step_out) * (size_t)n_cars) == hipSuccess;
        b->dOutActive = b->dOut;
    }
#ifdef PDB_STAMPS
    if (ok) {
        ok = ok && hipMalloc(&b->dStamps, sizeof(unsigned long long) * 64 * (size_t)n_cars) == hipSuccess;
        ok = ok && hipMemset(b->dStamps, 0, sizeof(unsigned long long) * 64 * (size_t)n_cars) == hipSuccess;
        b->K.stamps = b->dStamps; b->K.stampCars = n_cars;
        ok = ok && hipMemcpy(b->dK, &b->K, sizeof(DevConst), hipMemcpyHostToDevice) == hipSuccess;
    }
#endif
    if (!ok) { pdb::setError("pdb_create: HIP allocation / upload failed"); pdb_destroy(b); return nullptr; }
    try {
        pdb::TrackView tv(b->track.data());
        pdb::initialState(b->params, tv, b->resetTemplate);
    } catch (const std::exception& e) { pdb::setError(e.what()); pdb_destroy(b); return nullptr; }
    if (pdb_set_state_all(b, &b->resetTemplate) != PDB_OK) { pdb_destroy(b); return nullptr; }
    if (hipMalloc(&b->dFresh, sizeof(pdb_dyn_state)) != hipSuccess || hipMemcpy(b->dFresh, &b->resetTemplate, sizeof(pdb_dyn_state), hipMemcpyHostToDevice) != hipSuccess) {
        pdb::setError("pdb_create: HIP allocation / upload failed"); pdb_destroy(b); return nullptr;
    }
    b->K.freshState = b->dFresh;
    if (pushK(b, b->stream, false) != PDB_OK) { pdb_destroy(b); return nullptr; }
    ensureSnap(b);
    if (passNeeded(b, b->params) && !b->dSnap) { pdb_destroy(b); return nullptr; }   // (ensureSnap left the message)
    return b;
}

void pdb_destroy(pdb_batch* b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    commFree(b);
    if (b->graphExec) (void)hipGraphExecDestroy(b->graphExec);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    (void)hipFree(b->dStates); (void)hipFree(b->dActions); (void)hipFree(b->dOut); (void)hipFree(b->dCarStates); (void)hipFree(b->dParams); (void)hipFree(b->dK); (void)hipFree(b->dTrack); (void)hipFree(b->dContacts); (void)hipFree(b->dSlip); (void)hipFree(b->dFresh); for (int p = 0; p < PDB_MAX_PARTS; ++p) (void)hipFree(b->dPartFresh[p]); (void)hipFree(b->dResetMask); (void)hipFree(b->dResetScratch);
    if (b->hHint) (void)hipHostFree(b->hHint);
    if (b->hActions) (void)hipHostFree(b->hActions);
    if (b->hOut) (void)hipHostFree(b->hOut);
    for (int q = 0; q <= PDB_MAX_PARTS; ++q) (void)hipFree(b->dQueue[q]);
    for (int q = 0; q <= PDB_MAX_PARTS; ++q) for (hipEvent_t e : b->samples[q].ev) if (e) (void)hipEventDestroy(e);
    (void)hipFree(b->dSnap);
    (void)hipFree(b->dLaneTunes);
    (void)hipFree(b->dLawBias);
    (void)hipFree(b->dLaneSetups);
    (void)hipFree(b->dHold);
    for (int q = 0; q < PDB_MAX_PARTS; ++q) { (void)hipFree(b->dPartParams[q]); (void)hipFree(b->dPartK[q]); }
    if (b->ev0) (void)hipEventDestroy(b->ev0);
    if (b->ev1) (void)hipEventDestroy(b->ev1);
    if (b->tev0) (void)hipEventDestroy(b->tev0);
    if (b->tev1) (void)hipEventDestroy(b->tev1);
    for (int p = 0; p < PDB_MAX_PARTS; ++p) {
        if (b->partStream[p]) { (void)hipStreamSynchronize(b->partStream[p]); if (!(p == 0 && b->part0OnOwn)) (void)hipStreamDestroy(b->partStream[p]); }
        if (b->partEnd[p]) (void)hipEventDestroy(b->partEnd[p]);
        if (b->partStart[p]) (void)hipEventDestroy(b->partStart[p]);
    }
    if (b->partFork) (void)hipEventDestroy(b->partFork);
    if (b->stream && b->ownStream) (void)hipStreamDestroy(b->stream);
    delete b;
}

int pdb_num_cars(const pdb_batch* b) { return b ? b->n : 0; }

int pdb_set_state_all(pdb_batch* b, const pdb_dyn_state* state) {
    if (!b || !state) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    if (int rcj = joinParts(b)) return rcj;
    wideContactPasses(b);
    std::vector<pdb_dyn_state> tmp((size_t)b->n, *state);
    HIPCHK(hipMemcpyAsync(b->dStates, tmp.data(), sizeof(pdb_dyn_state) * (size_t)b->n, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return PDB_OK;
}
int pdb_set_state(pdb_batch* b, int first, int count, const pdb_dyn_state* states) {
    if (!b || !states || first < 0 || count < 0 || first + count > b->n) { pdb::setError("bad range"); return PDB_ERR_ARG; }
    if (int rcj = joinParts(b)) return rcj;
    wideContactPasses(b);
    HIPCHK(hipMemcpyAsync(b->dStates + first, states, sizeof(pdb_dyn_state) * (size_t)count, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return PDB_OK;
}
int pdb_get_state(pdb_batch* b, int first, int count, pdb_dyn_state* states) {
    if (!b || !states || first < 0 || count < 0 || first + count > b->n) { pdb::setError("bad range"); return PDB_ERR_ARG; }
    if (int rcj = joinParts(b)) return rcj;
    HIPCHK(hipMemcpyAsync(states, b->dStates + first, sizeof(pdb_dyn_state) * (size_t)count, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return PDB_OK;
}

int pdb_get_contacts(pdb_batch* b, int first, int count, pdb_contact* out) {
    if (!b || !out || first < 0 || count < 0 || first + count > b->n) { pdb::setError("bad range"); return PDB_ERR_ARG; }
    if (int rcj = joinParts(b)) return rcj;
    HIPCHK(hipMemcpyAsync(out, b->dContacts + (size_t)first * PDB_MAX_CONTACTS, sizeof(pdb_contact) * PDB_MAX_CONTACTS * (size_t)count, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return PDB_OK;
}
int pdb_set_contacts(pdb_batch* b, int first, int count, const pdb_contact* in) {
    if (!b || !in || first < 0 || count < 0 || first + count > b->n) { pdb::setError("bad range"); return PDB_ERR_ARG; }
    if (int rcj = joinParts(b)) return rcj;
    wideContactPasses(b);
    HIPCHK(hipMemcpyAsync(b->dContacts + (size_t)first * PDB_MAX_CONTACTS, in, sizeof(pdb_contact) * PDB_MAX_CONTACTS * (size_t)count, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return PDB_OK;
}

// Car::teleportByMode for the cars whose mask byte is non-zero (mask == nullptr: every car), one thread per car on the records in
// HBM; mode >= 0: that mode for every masked car, mode < 0: the byte is 1 + mode.  clear: zero the consumed bytes.
extern "C" __global__ void pdb_reset_kernel(pdb_dyn_state* __restrict__ states, uint8_t* __restrict__ mask, const pdb_car_params* __restrict__ Pp,
                                            const uint8_t* __restrict__ trackBlob, int n, int mode, int clear) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const int rq = mask ? (int)mask[i] : 1;
    if (rq == 0) return;
    k33::DevTrack T;
    T.h = reinterpret_cast<const pdb_track_header*>(trackBlob);
    T.surfaces = reinterpret_cast<const pdb_surface*>(trackBlob + T.h->offSurfaces);
    T.tris = reinterpret_cast<const float*>(trackBlob + T.h->offTris);
    T.fat = reinterpret_cast<const float*>(trackBlob + T.h->offFat);
    T.fatDist = reinterpret_cast<const float*>(trackBlob + T.h->offFatDist);
    T.nodes = reinterpret_cast<const float*>(trackBlob + T.h->offNodes);
    T.nodeDist = reinterpret_cast<const float*>(trackBlob + T.h->offNodeDist);
    k33::teleportByModeDev(*Pp, T, mode >= 0 ? mode : rq - 1, states + i);
    if (mask && clear) mask[i] = 0;
}
// env mode's per-car episode sums (cumulative reward, step count, the pending-reset flag) cleared on the device: the env's reset()
// without moving a record over PCIe (mask == nullptr: every car)
extern "C" __global__ void pdb_clear_episodes_kernel(pdb_dyn_state* __restrict__ states, const uint8_t* __restrict__ mask, int n) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n || (mask && !mask[i])) return;
    states[i].envTotalReward = 0.0; states[i].envPending = 0; states[i].envStepId = 0;
}
// After anything that puts cars down or rewrites their records (resets, teleports through the mask, pdb_set_state, pdb_set_contacts): the contact pass is
// launched wide for the next PDB_BURST_TICKS ticks of every launch site, whatever the hint says.  Cars just put down go through the pass -- all of them
// in the same ticks -- and the hint is both late (written at a pass's END) and read early (the host enqueues ticks ahead): with two wide ticks the
// third and fourth pass after a 16384-car reset still ran 20 ms each in ONE workgroup (profiles/r04: tools/contact_trace.py), eight cover the lag.
#define PDB_BURST_TICKS 8
static void wideContactPasses(pdb_batch* b) { for (int q = 0; q <= PDB_MAX_PARTS; ++q) b->burst[q] = PDB_BURST_TICKS; }
static int resetLaunch(pdb_batch* b, uint8_t* dMask, int mode, int clear) {
    if (int rcj = joinParts(b)) return rcj;
    wideContactPasses(b);
    bool any = false;
    for (int p = 0; p < b->parts; ++p) any = any || b->partHas[p];
    if (!any) hipLaunchKernelGGL(pdb_reset_kernel, dim3((b->n + 63) / 64), dim3(64), 0, b->stream, b->dStates, dMask, b->dParams, b->dTrack, b->n, mode, clear);
    else for (int p = 0; p < b->parts; ++p) {   // each partition's cars with the partition's own block (fuel, ride height ...)
        const int c0 = partFirst(b, p), c1 = partFirst(b, p + 1);
        if (c1 <= c0) continue;
        hipLaunchKernelGGL(pdb_reset_kernel, dim3((c1 - c0 + 63) / 64), dim3(64), 0, b->stream, b->dStates + c0, dMask ? dMask + c0 : nullptr,
                           b->partHas[p] ? b->dPartParams[p] : b->dParams, b->dTrack, c1 - c0, mode, clear);
    }
    LAUNCHCHK(b);
    return PDB_OK;
}
int pdb_reset_mode(pdb_batch* b, const uint8_t* mask, int mode) {
    if (!b || mode < 0 || mode > 2) { pdb::setError("pdb_reset_mode: bad argument"); return PDB_ERR_ARG; }
    // teleportCarByMode: Car::teleportToSpline + Car::reset applied to each masked car's record, on the device (the host hands
    // over the mask only)
    if (!mask) { int rc = resetLaunch(b, nullptr, mode, 0); if (rc != PDB_OK) return rc; HIPCHK(hipStreamSynchronize(b->stream)); return PDB_OK; }
    if (int rcj = joinParts(b)) return rcj;
    if (!b->dResetScratch) HIPCHK(hipMalloc(&b->dResetScratch, (size_t)b->n));
    HIPCHK(hipMemcpyAsync(b->dResetScratch, mask, (size_t)b->n, hipMemcpyHostToDevice, b->stream));
    int rc = resetLaunch(b, b->dResetScratch, mode, 0);
    if (rc != PDB_OK) return rc;
    HIPCHK(hipStreamSynchronize(b->stream));   // the caller's mask may be reused at once
    return PDB_OK;
}
int pdb_reset(pdb_batch* b, const uint8_t* mask) { return pdb_reset_mode(b, mask, 0); }
int pdb_reset_device(pdb_batch* b, const uint8_t* device_mask, int mode) {
    if (!b || !device_mask || mode < 0 || mode > 2) { pdb::setError("pdb_reset_device: bad argument"); return PDB_ERR_ARG; }
    b->batchDirty = true;
    return resetLaunch(b, const_cast<uint8_t*>(device_mask), mode, 0);   // asynchronous on the batch's stream: no host round trip
}
int pdb_clear_episodes(pdb_batch* b, const uint8_t* device_mask) {
    if (!b) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    if (int rcj = joinParts(b)) return rcj;
    hipLaunchKernelGGL(pdb_clear_episodes_kernel, dim3((b->n + 255) / 256), dim3(256), 0, b->stream, b->dStates, device_mask, b->n);
    LAUNCHCHK(b);
    b->batchDirty = true;
    return PDB_OK;
}
uint8_t* pdb_reset_mask_device(pdb_batch* b) {
    if (!b) return nullptr;
    b->resetMaskArmed = true;
    ensureSnap(b);
    if (!b->dSnap) { b->resetMaskArmed = false; return nullptr; }   // (out of device memory: the message is set)
    if (b->graphExec) { (void)hipGraphExecDestroy(b->graphExec); b->graphExec = nullptr; }   // captured launches carry the old argument
    return b->dResetMask;
}
int pdb_set_stuck_timeout(pdb_batch* b, double seconds) {
    if (!b || !(seconds >= 0.0)) { pdb::setError("pdb_set_stuck_timeout: bad argument"); return PDB_ERR_ARG; }
    if (int rcj = joinParts(b)) return rcj;
    for (int p = 0; p < b->parts; ++p) if (b->partStream[p]) HIPCHK(hipStreamSynchronize(b->partStream[p]));
    HIPCHK(hipStreamSynchronize(b->stream));
    b->K.stuckTimeout = seconds;
    if (int rck = pushK(b, b->stream, false)) return rck;
    return PDB_OK;
}
int pdb_set_partition_params(pdb_batch* b, int part, const pdb_car_params* params) {
    if (!b || part < 0 || part >= b->parts || b->parts < 2) { pdb::setError("pdb_set_partition_params: no such partition (pdb_set_partitions first)"); return PDB_ERR_ARG; }
    if (b->dLaneSetups) { pdb::setError("pdb_set_partition_params: the per-lane setup table exists and holds the lanes' RESOLVED values (install the partitions' blocks before pdb_set_lane_setups)"); return PDB_ERR_ARG; }
    if (int rcj = joinParts(b)) return rcj;
    for (int p = 0; p < PDB_MAX_PARTS; ++p) if (b->partStream[p]) HIPCHK(hipStreamSynchronize(b->partStream[p]));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (!params) {
        b->partHas[part] = false;
        if (b->graphExec) { (void)hipGraphExecDestroy(b->graphExec); b->graphExec = nullptr; }   // the captured launches carry the partitions' own blocks
        return pushK(b, b->stream, false);   // the partition's constants block goes back to the batch's values
    }
    const pdb_car_params& A = b->params;
    bool same = params->numBodies == A.numBodies && params->numJoints == A.numJoints && params->numRows == A.numRows;
    for (int j = 0; same && j < A.numJoints; ++j) same = params->joints[j].type == A.joints[j].type && params->joints[j].b0 == A.joints[j].b0 && params->joints[j].b1 == A.joints[j].b1;
#ifdef PDB_FAST_BUILD
    if (params->numCtrlStages != 0 || params->hasBrakeTemps != 0) { pdb::setError("pdb_set_partition_params: this development build (PDB_FAST_BUILD) holds no controller kernels"); return PDB_ERR_ARG; }
#endif
    if (!same) { pdb::setError("pdb_set_partition_params: the block's rigid-body topology differs from the batch's (one kernel variant per batch)"); return PDB_ERR_ARG; }
    if (!b->dPartParams[part]) HIPCHK(hipMalloc(&b->dPartParams[part], sizeof(pdb_car_params)));
    if (!b->dPartK[part]) HIPCHK(hipMalloc(&b->dPartK[part], sizeof(DevConst)));
    b->partParams[part] = *params;
    b->partHas[part] = true;
    ensureSnap(b);
    if (passNeeded(b, *params) && !b->dSnap) { b->partHas[part] = false; return PDB_ERR_HIP; }
    HIPCHK(hipMemcpy(b->dPartParams[part], params, sizeof(pdb_car_params), hipMemcpyHostToDevice));
    {   // the record a faulted car of this partition is re-created from (env mode): this block's, not the batch's
        pdb_dyn_state fresh;
        try {
            pdb::TrackView tv(b->track.data());
            pdb::initialState(*params, tv, fresh);
        } catch (const std::exception& e) { pdb::setError(e.what()); b->partHas[part] = false; return PDB_ERR_IO; }
        if (!b->dPartFresh[part]) HIPCHK(hipMalloc(&b->dPartFresh[part], sizeof(pdb_dyn_state)));
        HIPCHK(hipMemcpy(b->dPartFresh[part], &fresh, sizeof(pdb_dyn_state), hipMemcpyHostToDevice));
    }
    if (b->graphExec) { (void)hipGraphExecDestroy(b->graphExec); b->graphExec = nullptr; }
    return pushK(b, b->stream, false);
}
// the rows of the lanes [first, first + count) from their own blocks: the partition's where it has one, else the batch's
static int laneSetupDefaults(pdb_batch* b, int first, int count) {
    std::vector<pdb_lane_setup> rows((size_t)count);
    pdb_lane_setup own, part[PDB_MAX_PARTS];
    (void)pdb_lane_setup_from_params(&b->params, &own);
    for (int p = 0; p < PDB_MAX_PARTS; ++p) if (b->partHas[p]) (void)pdb_lane_setup_from_params(&b->partParams[p], &part[p]);
    for (int i = 0; i < count; ++i) {
        int p = 0;
        while (p + 1 < b->parts && first + i >= partFirst(b, p + 1)) ++p;
        rows[(size_t)i] = (b->parts > 1 && p < PDB_MAX_PARTS && b->partHas[p]) ? part[p] : own;
    }
    HIPCHK(hipMemcpy(b->dLaneSetups + first, rows.data(), sizeof(pdb_lane_setup) * (size_t)count, hipMemcpyHostToDevice));
    return PDB_OK;
}
int pdb_set_lane_tunes(pdb_batch* b, int first, int count, const pdb_lane_tune* rows) {
    if (!b || first < 0 || count < 0 || first + count > b->n) { pdb::setError("pdb_set_lane_tunes: bad range"); return PDB_ERR_ARG; }
    if (int rcj = joinParts(b)) return rcj;
    for (int p = 0; p < PDB_MAX_PARTS; ++p) if (b->partStream[p]) HIPCHK(hipStreamSynchronize(b->partStream[p]));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (!b->dLaneTunes) {
        if (!rows) return PDB_OK;   // nothing installed, nothing to take back
        HIPCHK(hipMalloc(&b->dLaneTunes, sizeof(pdb_lane_tune) * (size_t)b->n));
        HIPCHK(hipMemset(b->dLaneTunes, 0, sizeof(pdb_lane_tune) * (size_t)b->n));   // valid = 0 everywhere
        b->K.laneTunes = b->dLaneTunes;
        if (int rck = pushK(b, b->stream, false)) return rck;
    }
    if (count == 0) return PDB_OK;
    if (rows) {
        for (int i = 0; i < count; ++i) if (rows[i].valid != 0 && rows[i].valid != 1) { pdb::setError("pdb_set_lane_tunes: a row's valid field is neither 0 nor 1 (fill rows with pdb_lane_tune_from_params)"); return PDB_ERR_ARG; }
        HIPCHK(hipMemcpy(b->dLaneTunes + first, rows, sizeof(pdb_lane_tune) * (size_t)count, hipMemcpyHostToDevice));
    } else HIPCHK(hipMemset(b->dLaneTunes + first, 0, sizeof(pdb_lane_tune) * (size_t)count));
    return PDB_OK;
}
int pdb_set_lane_setups(pdb_batch* b, int first, int count, const pdb_lane_setup* rows) {
    if (!b || first < 0 || count < 0 || first + count > b->n) { pdb::setError("pdb_set_lane_setups: bad range"); return PDB_ERR_ARG; }
    if (int rcj = joinParts(b)) return rcj;
    for (int p = 0; p < PDB_MAX_PARTS; ++p) if (b->partStream[p]) HIPCHK(hipStreamSynchronize(b->partStream[p]));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (!b->dLaneSetups) {
        if (!rows) return PDB_OK;   // nothing installed, nothing to take back
        HIPCHK(hipMalloc(&b->dLaneSetups, sizeof(pdb_lane_setup) * (size_t)b->n));
        // a table that did not get its rows or did not reach the constants blocks must not exist: launchTick picks the table's kernels by dLaneSetups alone
        auto undo = [&](int rc) { (void)hipFree(b->dLaneSetups); b->dLaneSetups = nullptr; b->K.laneSetups = nullptr; return rc; };
        if (int rcd = laneSetupDefaults(b, 0, b->n)) return undo(rcd);
        b->K.laneSetups = b->dLaneSetups;
        if (b->graphExec) { (void)hipGraphExecDestroy(b->graphExec); b->graphExec = nullptr; b->graphTicks = 0; }   // recorded launches are the other kernel pair's
        if (int rck = pushK(b, b->stream, false)) { (void)undo(rck); (void)pushK(b, b->stream, false); return rck; }
    }
    if (count == 0) return PDB_OK;
    if (rows) HIPCHK(hipMemcpy(b->dLaneSetups + first, rows, sizeof(pdb_lane_setup) * (size_t)count, hipMemcpyHostToDevice));
    else if (int rcd = laneSetupDefaults(b, first, count)) return rcd;
    return PDB_OK;
}
// The device law: the next tick's steer / throttle components worked out by the tick's own launches from the observation row they write (step_kernel.hip.inc, phase 7),
// a[c] = bias[c] + sum_k obs[k] * weights[k][c] -- scripted input streams (a per-car, per-tick table of biases with a period) and linear feedback laws of the observation
// (projectd_env.py's 24 slots) without a policy launch between two ticks.  The action buffer stays the caller's to read (pdb_actions_device) and to overwrite.
int pdb_set_law(pdb_batch* b, const float* weights, const float* bias0, const float* table, int period, int table_on_device) {
    if (!b) { pdb::setError("pdb_set_law: no batch"); return PDB_ERR_ARG; }
    if (weights && b->K.actionMode == PDB_ACTION_FULL) { pdb::setError("pdb_set_law: the law writes two action components; this batch takes all eight controls"); return PDB_ERR_ARG; }
    if (weights && table && period < 1) { pdb::setError("pdb_set_law: a bias table needs a period of at least one row"); return PDB_ERR_ARG; }
    if (int rcj = joinParts(b)) return rcj;
    for (int p = 0; p < PDB_MAX_PARTS; ++p) if (b->partStream[p]) HIPCHK(hipStreamSynchronize(b->partStream[p]));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (b->graphExec) { (void)hipGraphExecDestroy(b->graphExec); b->graphExec = nullptr; b->graphTicks = 0; }
    (void)hipFree(b->dLawBias); b->dLawBias = nullptr;
    b->K.lawPeriod = 0; b->K.lawStride = 0; b->K.lawBias = nullptr; b->K.lawBias0[0] = 0.0f; b->K.lawBias0[1] = 0.0f; memset(b->K.lawW, 0, sizeof(b->K.lawW));
    if (weights) {
        memcpy(b->K.lawW, weights, sizeof(b->K.lawW));
        if (bias0) { b->K.lawBias0[0] = bias0[0]; b->K.lawBias0[1] = bias0[1]; }
        b->K.lawPeriod = table ? period : 1; b->K.lawStride = b->n;
        if (table && table_on_device) b->K.lawBias = table;
        else if (table) {
            const size_t bytes = sizeof(float) * 2 * (size_t)b->n * (size_t)period;
            if (hipMalloc(&b->dLawBias, bytes) != hipSuccess) { b->dLawBias = nullptr; b->K.lawPeriod = 0; (void)pushK(b, b->stream, false); pdb::setError("pdb_set_law: out of device memory for the bias table"); return PDB_ERR_HIP; }
            HIPCHK(hipMemcpy(b->dLawBias, table, bytes, hipMemcpyHostToDevice));
            b->K.lawBias = b->dLawBias;
        }
    }
    return pushK(b, b->stream, false);
}
// Multi-car simulators (reference Sim/Simulator.cpp:59-60,112-153: 1..100 cars share a Simulator; cfg/sim.ini ships MAX_CARS = 2): the batch's cars form worlds of
// `cars_per_world` consecutive lanes.  What couples the cars of a world on this path is the slipstream (Car::updateAirPressure, Car.cpp:557-585: the air a car meets is
// thinned by the wakes of the others); body contacts between cars are not built.
int pdb_set_world_size(pdb_batch* b, int cars_per_world) {
    if (!b || cars_per_world < 1 || cars_per_world > 100 || b->n % cars_per_world != 0) { pdb::setError("pdb_set_world_size: 1..100 cars per world, a divisor of the batch's car count"); return PDB_ERR_ARG; }
    if (int rcj = joinParts(b)) return rcj;
    for (int p = 0; p < PDB_MAX_PARTS; ++p) if (b->partStream[p]) HIPCHK(hipStreamSynchronize(b->partStream[p]));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (cars_per_world > 1 && !b->dSlip) {
        HIPCHK(hipMalloc(&b->dSlip, sizeof(pdb_slip_state) * 2 * (size_t)b->n));
        HIPCHK(hipMemset(b->dSlip, 0, sizeof(pdb_slip_state) * 2 * (size_t)b->n));   // a new car has no wake (SlipStream.h: length 0) until its first tick ends
    }
    if (b->parts > 1 && cars_per_world != (b->K.worldSize > 1 ? b->K.worldSize : 1)) commFree(b);   // the partitions' cuts move to whole worlds: their communicators' ranges are the old cut's (pdb_comm_init again)
    b->K.worldSize = cars_per_world; b->K.slipStride = b->n; b->K.slip = cars_per_world > 1 ? b->dSlip : nullptr;
    if (b->graphExec) { (void)hipGraphExecDestroy(b->graphExec); b->graphExec = nullptr; b->graphTicks = 0; }
    return pushK(b, b->stream, false);   // (the partitions' boundaries move to whole worlds: partFirst)
}
// the slipstreams as state (a snapshot that must replay bit for bit carries them next to the records): both buffers, [0][count] then [1][count] -- a car's next tick reads
// the one of its frame counter's parity (pdb_dyn_state.simFrame & 1)
int pdb_get_slipstreams(pdb_batch* b, int first, int count, pdb_slip_state* out) {
    if (!b || !out || first < 0 || count < 0 || first + count > b->n || !b->dSlip) { pdb::setError("pdb_get_slipstreams: bad range, or not a batch of multi-car simulators"); return PDB_ERR_ARG; }
    if (int rcj = joinParts(b)) return rcj;
    HIPCHK(hipStreamSynchronize(b->stream));
    for (int k = 0; k < 2; ++k) HIPCHK(hipMemcpy(out + (size_t)k * count, b->dSlip + (size_t)k * b->n + first, sizeof(pdb_slip_state) * (size_t)count, hipMemcpyDeviceToHost));
    return PDB_OK;
}
int pdb_set_slipstreams(pdb_batch* b, int first, int count, const pdb_slip_state* in) {
    if (!b || !in || first < 0 || count < 0 || first + count > b->n || !b->dSlip) { pdb::setError("pdb_set_slipstreams: bad range, or not a batch of multi-car simulators"); return PDB_ERR_ARG; }
    if (int rcj = joinParts(b)) return rcj;
    HIPCHK(hipStreamSynchronize(b->stream));
    for (int k = 0; k < 2; ++k) HIPCHK(hipMemcpy(b->dSlip + (size_t)k * b->n + first, in + (size_t)k * count, sizeof(pdb_slip_state) * (size_t)count, hipMemcpyHostToDevice));
    return PDB_OK;
}
int pdb_set_env(pdb_batch* b, const pdb_env_config* cfg) {
    if (!b || !cfg || cfg->teleport_mode < 0 || cfg->teleport_mode > 2) { pdb::setError("pdb_set_env: bad argument"); return PDB_ERR_ARG; }
    if (int rcj = joinParts(b)) return rcj;
    for (int p = 0; p < b->parts; ++p) if (b->partStream[p]) HIPCHK(hipStreamSynchronize(b->partStream[p]));
    HIPCHK(hipStreamSynchronize(b->stream));
    DevConst& K = b->K;
    K.envMode = cfg->enabled ? 1 : 0;
    K.envTermHit = cfg->terminate_on_hit ? 1 : 0; K.envTermOff = cfg->terminate_off_track ? 1 : 0; K.envTermStuck = cfg->terminate_when_stuck ? 1 : 0;
    K.envHitPenalty = cfg->hit_penalty; K.envOffPenalty = cfg->off_track_penalty; K.envStuckPenalty = cfg->stuck_penalty; K.envLowReward = cfg->low_reward;
    K.envTeleportOnReset = cfg->teleport_on_reset ? 1 : 0; K.envTeleportMode = cfg->teleport_mode;
    if (int rck = pushK(b, b->stream, false)) return rck;
    return PDB_OK;
}
int pdb_set_seed(pdb_batch* b, const uint32_t* seeds) {
    if (!b || !seeds) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    std::vector<pdb_dyn_state> st((size_t)b->n);
    int rc = pdb_get_state(b, 0, b->n, st.data());
    if (rc != PDB_OK) return rc;
    for (int i = 0; i < b->n; ++i) st[(size_t)i].randState = (int32_t)seeds[i];   // setSeed = srand (PyProjectD.cpp:50-53), one C runtime per car
    return pdb_set_state(b, 0, b->n, st.data());
}

float* pdb_actions_device(pdb_batch* b) { return b ? b->dActions : nullptr; }
pdb_step_out* pdb_out_device(pdb_batch* b) { return b ? b->dOutActive : nullptr; }
int pdb_set_out_device(pdb_batch* b, pdb_step_out* out) {
    if (!b) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    pdb_step_out* p = out ? out : b->dOut;
    if (p != b->dOutActive && b->graphExec) { (void)hipGraphExecDestroy(b->graphExec); b->graphExec = nullptr; }   // the captured launches carry the old pointer
    b->dOutActive = p;
    return PDB_OK;
}
void* pdb_stream(pdb_batch* b) { return b ? (void*)b->stream : nullptr; }

int pdb_step(pdb_batch* b, float dt) {
    if (!b) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    HIPCHK(hipEventRecord(b->ev0, b->stream));
    int rc = launch(b, dt, false);
    if (rc != PDB_OK) return rc;
    HIPCHK(hipEventRecord(b->ev1, b->stream));
    HIPCHK(hipEventSynchronize(b->ev1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, b->ev0, b->ev1));
    b->kernelMs += ms; b->kernelLaunches += 1;
    return PDB_OK;
}

int pdb_step_n(pdb_batch* b, float dt, int n) {
    if (!b || n <= 0) { pdb::setError("bad argument"); return PDB_ERR_ARG; }
    if (int rcj = joinParts(b)) return rcj;
    if (n == 1) return pdb_step(b, dt);
    // the constants are made current before any graph launch, not only when the graph is rebuilt (a pdb_step_host in between
    // leaves wantCarState = 1 behind); the capture must not contain the H2D copy
    if (b->K.dt != dt || b->K.wantCarState != 0) {
        b->K.dt = dt; b->K.fps = 1.0f / dt; b->K.dtD = (dt == (float)(1.0 / 333.0)) ? (1.0 / 333.0) : (double)dt; b->K.wantCarState = 0;
        if (int rck = pushK(b, b->stream, true)) return rck;
        HIPCHK(hipStreamSynchronize(b->stream));
    }
    if (b->stream == nullptr) {   // the legacy default stream (a caller's pdb_set_stream) cannot be captured: n plain launches
        HIPCHK(hipEventRecord(b->ev0, b->stream));
        for (int i = 0; i < n; ++i) launchWhole(b, b->stream, b->dOutActive);
        LAUNCHCHK(b);
        HIPCHK(hipEventRecord(b->ev1, b->stream));
        HIPCHK(hipEventSynchronize(b->ev1));
        float ms0 = 0;
        HIPCHK(hipEventElapsedTime(&ms0, b->ev0, b->ev1));
        b->kernelMs += ms0; b->kernelLaunches += n;
        return PDB_OK;
    }
    if (!b->graphExec || b->graphTicks != n || b->graphDt != dt) {
        if (b->graphExec) { (void)hipGraphExecDestroy(b->graphExec); b->graphExec = nullptr; }
        hipGraph_t g = nullptr;
        HIPCHK(hipStreamBeginCapture(b->stream, hipStreamCaptureModeThreadLocal));
        b->capturing = true;
        for (int i = 0; i < n; ++i)
            launchWhole(b, b->stream, b->dOutActive);
        b->capturing = false;
        HIPCHK(hipStreamEndCapture(b->stream, &g));
        HIPCHK(hipGraphInstantiate(&b->graphExec, g, nullptr, nullptr, 0));
        (void)hipGraphDestroy(g);
        b->graphTicks = n; b->graphDt = dt;
    }
    HIPCHK(hipEventRecord(b->ev0, b->stream));
    HIPCHK(hipGraphLaunch(b->graphExec, b->stream));
    HIPCHK(hipEventRecord(b->ev1, b->stream));
    HIPCHK(hipEventSynchronize(b->ev1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, b->ev0, b->ev1));
    b->kernelMs += ms; b->kernelLaunches += n;
    return PDB_OK;
}

int pdb_step_async(pdb_batch* b, float dt) {
    if (!b) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    return launch(b, dt, false);
}

int pdb_set_stream(pdb_batch* b, void* hip_stream) {
    if (!b) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    if (int rcj = joinParts(b)) return rcj;
    HIPCHK(hipStreamSynchronize(b->stream));
    if (b->graphExec) { (void)hipGraphExecDestroy(b->graphExec); b->graphExec = nullptr; b->graphTicks = 0; }
    if (b->part0OnOwn) {   // the own stream goes: partition 0 gets one of its own (the count of the library's streams stays what it was)
        b->partStream[0] = nullptr; b->part0OnOwn = false;
        HIPCHK(hipStreamCreateWithFlags(&b->partStream[0], hipStreamNonBlocking));
    }
    if (b->ownStream) { (void)hipStreamDestroy(b->stream); b->ownStream = false; }
    b->stream = (hipStream_t)hip_stream;
    return PDB_OK;
}

// diagnostic: the number of cars the last contact pass of a launch site held (site = partition, PDB_MAX_PARTS = the batch's own stream)
int pdb_contact_pass_load(pdb_batch* b, int site) {
    if (!b || site < 0 || site > PDB_MAX_PARTS || !b->hHint) return -1;
    return *(volatile int*)(b->hHint + site);
}
int pdb_set_contact_grid(pdb_batch* b, int workgroups) {
    if (!b || workgroups < 0 || workgroups > 4096) { pdb::setError("pdb_set_contact_grid: 0 (adaptive) .. 4096 workgroups"); return PDB_ERR_ARG; }
    b->contactGrid = workgroups;
    if (b->graphExec) { (void)hipGraphExecDestroy(b->graphExec); b->graphExec = nullptr; }
    return PDB_OK;
}
int pdb_set_partitions(pdb_batch* b, int parts) {
    if (!b || parts < 1 || parts > PDB_MAX_PARTS) { pdb::setError("pdb_set_partitions: 1..4 parts"); return PDB_ERR_ARG; }
    if (b->dLaneSetups && parts != b->parts) for (int p = 0; p < PDB_MAX_PARTS; ++p) if (b->partHas[p]) {   // (the table's default rows were read out of the old cut's blocks)
        pdb::setError("pdb_set_partitions: the per-lane setup table holds rows resolved from the partitions' own car blocks: cut the batch before pdb_set_lane_setups"); return PDB_ERR_ARG; }
    HIPCHK(hipSetDevice(b->device));
    if (int rcj = joinParts(b)) return rcj;
    for (int p = 0; p < PDB_MAX_PARTS; ++p) if (b->partStream[p]) HIPCHK(hipStreamSynchronize(b->partStream[p]));   // the old cut's kernels
    HIPCHK(hipStreamSynchronize(b->stream));
    for (int p = 0; p < parts; ++p) {
        // A process has four hardware queues and the null stream holds one: a FOURTH stream of the library's would share a queue with one of the
        // others, and kernels of two streams in one queue run one after the other (two of three partitions at half speed: tools/region_ticks.py,
        // 2015 against 1285 us for twenty ticks of 4096 cars).  While the batch runs on the library's own stream, partition 0 runs on that stream too.
        if (p == 0 && !b->partStream[0] && b->ownStream) { b->partStream[0] = b->stream; b->part0OnOwn = true; }
        if (!b->partStream[p]) HIPCHK(hipStreamCreateWithFlags(&b->partStream[p], hipStreamNonBlocking));
        if (!b->partEnd[p]) HIPCHK(hipEventCreate(&b->partEnd[p]));
        if (!b->partStart[p]) HIPCHK(hipEventCreate(&b->partStart[p]));
    }
    if (!b->partFork) HIPCHK(hipEventCreateWithFlags(&b->partFork, hipEventDisableTiming));
    if (b->graphExec) { (void)hipGraphExecDestroy(b->graphExec); b->graphExec = nullptr; }   // captured launches follow the old cut
    if (parts != b->parts) commFree(b);   // the partitions' communicators belong to the old cut (pdb_comm_init again)
    b->parts = parts;
    if (parts > 1) for (int p = 0; p < parts; ++p) if (!b->dPartK[p]) HIPCHK(hipMalloc(&b->dPartK[p], sizeof(DevConst)));
    return pushK(b, b->stream, false);   // the partitions' constants blocks: the per-car tables start at each partition's first car
}

int pdb_step_ring(pdb_batch* b, float dt, int n_ticks, pdb_step_out* ring, int ring_slots, int first_slot, int join) {
    if (!b || n_ticks <= 0 || (ring && (ring_slots <= 0 || first_slot < 0))) { pdb::setError("pdb_step_ring: bad argument"); return PDB_ERR_ARG; }
    if (b->K.dt != dt || b->K.wantCarState != 0) {
        if (int rcj = joinParts(b)) return rcj;   // the constants change under kernels that may still read them: those first
        b->K.dt = dt; b->K.fps = 1.0f / dt; b->K.dtD = (dt == (float)(1.0 / 333.0)) ? (1.0 / 333.0) : (double)dt; b->K.wantCarState = 0;
        if (int rck = pushK(b, b->stream, true)) return rck;
        b->batchDirty = true;   // the copies are queued on the batch's stream: this ring forks behind them even when the caller asked for join bit 1 (no fork)
    }
    const bool forked = b->parts > 1 && b->partStream[0];
    // join bit 1 (PDB_RING_NO_FORK): the partitions are not held back behind what is queued on the batch's stream -- for a caller that orders its own dependencies
    // on the partitions' streams (sharding.TrajectoryGather: a ring is not rewritten before its previous gather has read it).  Without it every ring starts behind
    // the batch's stream, and a stream that shares that stream's hardware queue -- the gather's -- holds every partition up for as long as its kernel runs
    const bool noFork = (join & 2) != 0 && !b->batchDirty;
    join &= 1;
    if (forked && !noFork) HIPCHK(hipEventRecord(b->partFork, b->stream));
    if (forked && !noFork) b->batchDirty = false;
    // enqueued tick by tick across the partitions, not partition by partition: every range starts (and ends) within a few launches
    // of the others -- partition-major order left the last range idle for the first hundreds of microseconds of a short call and
    // alone on the GPU for the last ones
    const int np = forked ? b->parts : 1;
    for (int p = 0; p < np; ++p) {
        const int c0 = forked ? partFirst(b, p) : 0, c1 = forked ? partFirst(b, p + 1) : b->n;
        if (c1 <= c0) continue;
        hipStream_t st = forked ? b->partStream[p] : b->stream;
        if (forked && !noFork) HIPCHK(hipStreamWaitEvent(st, b->partFork, 0));
        if (forked && b->partMark) HIPCHK(hipEventRecord(b->partStart[p], st));
    }
    for (int i = 0; i < n_ticks; ++i) {
        pdb_step_out* out = ring ? ring + (size_t)((first_slot + i) % ring_slots) * (size_t)b->n : b->dOutActive;
        for (int p = 0; p < np; ++p) {
            const int c0 = forked ? partFirst(b, p) : 0, c1 = forked ? partFirst(b, p + 1) : b->n;
            if (c1 <= c0) continue;
            launchTick(b, forked ? b->partStream[p] : b->stream, c0, c1, out, forked ? p : PDB_MAX_PARTS);
        }
    }
    LAUNCHCHK(b);
    for (int p = 0; p < np; ++p) {
        const int c0 = forked ? partFirst(b, p) : 0, c1 = forked ? partFirst(b, p + 1) : b->n;
        if (c1 <= c0) continue;
        if (forked) { HIPCHK(hipEventRecord(b->partEnd[p], b->partStream[p])); b->partEndStale[p] = false; if (join) HIPCHK(hipStreamWaitEvent(b->stream, b->partEnd[p], 0)); }
    }
    if (forked) { b->partMark = false; if (!join) b->partDirty = true; }
    return PDB_OK;
}
// one tick of one partition on its own stream, nothing forked or joined: for callers that keep a whole per-partition loop
// (kernel + their own work on pdb_partition_stream) running independently of the other partitions
int pdb_step_partition(pdb_batch* b, float dt, int part, pdb_step_out* out) {
    if (!b || part < 0 || part >= b->parts || b->parts < 2 || !b->partStream[part]) { pdb::setError("pdb_step_partition: no such partition"); return PDB_ERR_ARG; }
    if (b->K.dt != dt || b->K.wantCarState != 0) {   // constants change: everything in flight first
        if (int rcj = joinParts(b)) return rcj;
        b->K.dt = dt; b->K.fps = 1.0f / dt; b->K.dtD = (dt == (float)(1.0 / 333.0)) ? (1.0 / 333.0) : (double)dt; b->K.wantCarState = 0;
        if (int rck = pushK(b, b->stream, true)) return rck;
        HIPCHK(hipStreamSynchronize(b->stream));
    }
    const int c0 = partFirst(b, part), c1 = partFirst(b, part + 1);
    if (c1 <= c0) return PDB_OK;
    hipStream_t st = b->partStream[part];
    if (b->batchDirty) {   // asynchronous work on the batch's stream since the partitions last forked from it (pdb_reset_device, mask uploads): after it
        HIPCHK(hipEventRecord(b->partFork, b->stream));
        for (int p = 0; p < b->parts; ++p) HIPCHK(hipStreamWaitEvent(b->partStream[p], b->partFork, 0));
        b->batchDirty = false;
    }
    if (b->partMark) { for (int p = 0; p < b->parts; ++p) HIPCHK(hipEventRecord(b->partStart[p], b->partStream[p])); b->partMark = false; }
    pdb_step_out* o = out ? out : b->dOutActive;
    launchTick(b, st, c0, c1, o, part);
    LAUNCHCHK(b);
    b->partEndStale[part] = true;   // (recorded when somebody waits for it: freshEnd)
    b->partDirty = true;
    return PDB_OK;
}
static int hostMirrors(pdb_batch* b) {
    if (!b->hActions) { HIPCHK(hipHostMalloc((void**)&b->hActions, sizeof(float) * b->actionStride * (size_t)b->n, hipHostMallocDefault)); memset(b->hActions, 0, sizeof(float) * b->actionStride * (size_t)b->n); }
    if (!b->hOut) { HIPCHK(hipHostMalloc((void**)&b->hOut, sizeof(pdb_step_out) * (size_t)b->n, hipHostMallocDefault)); memset(b->hOut, 0, sizeof(pdb_step_out) * (size_t)b->n); }
    return PDB_OK;
}
float* pdb_host_actions(pdb_batch* b) { return (b && hostMirrors(b) == PDB_OK) ? b->hActions : nullptr; }
pdb_step_out* pdb_host_out(pdb_batch* b) { return (b && hostMirrors(b) == PDB_OK) ? b->hOut : nullptr; }
// actions up, one tick, outputs down -- for one partition, on its stream, nothing waited for
int pdb_step_host_partition(pdb_batch* b, float dt, int part) {
    if (!b || part < 0 || part >= b->parts || b->parts < 2 || !b->partStream[part]) { pdb::setError("pdb_step_host_partition: no such partition (pdb_set_partitions first)"); return PDB_ERR_ARG; }
    if (int rcm = hostMirrors(b)) return rcm;
    const int c0 = partFirst(b, part), c1 = partFirst(b, part + 1);
    if (c1 <= c0) return PDB_OK;
    hipStream_t st = b->partStream[part];
    HIPCHK(hipMemcpyAsync(b->dActions + (size_t)c0 * b->actionStride, b->hActions + (size_t)c0 * b->actionStride, sizeof(float) * b->actionStride * (size_t)(c1 - c0), hipMemcpyHostToDevice, st));
    int rc = pdb_step_partition(b, dt, part, nullptr);
    if (rc != PDB_OK) return rc;
    HIPCHK(hipMemcpyAsync(b->hOut + c0, b->dOutActive + c0, sizeof(pdb_step_out) * (size_t)(c1 - c0), hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(b->partEnd[part], st));   // the partition's end now includes the download
    b->partEndStale[part] = false;
    return PDB_OK;
}
int pdb_wait_host_partition(pdb_batch* b, int part) {
    if (!b || part < 0 || part >= b->parts || b->parts < 2 || !b->partEnd[part]) { pdb::setError("pdb_wait_host_partition: no such partition"); return PDB_ERR_ARG; }
    if (int rcf = freshEnd(b, part)) return rcf;
    HIPCHK(hipEventSynchronize(b->partEnd[part]));
    return PDB_OK;
}
void* pdb_partition_stream(pdb_batch* b, int part) { return (b && part >= 0 && part < b->parts && b->parts > 1) ? (void*)b->partStream[part] : nullptr; }
int pdb_partition_range(pdb_batch* b, int part, int* first, int* count) {
    if (!b || part < 0 || part >= b->parts || !first || !count) { pdb::setError("pdb_partition_range: bad argument"); return PDB_ERR_ARG; }
    *first = partFirst(b, part); *count = partFirst(b, part + 1) - partFirst(b, part);
    return PDB_OK;
}
// ---- the learner exchange of SURVEY 8e issued by the library: per partition and tick, on the partition's own stream and its own RCCL
// communicator, scatter of the partition's action rows from rank 0 -> the partition's tick -> all-gather of its output rows.  Three enqueues
// from C per partition and tick, nothing of the host's in between (through torch.distributed the same three steps cost six calls of ~25 us each,
// more than the tick takes on the GPU).  RCCL is taken from the process at run time (dlopen by soname: in a torch process that is the RCCL torch
// itself uses, otherwise /opt/rocm's), so the library carries no link-time dependency on it and loads on machines without it.
struct RcclApi {
    void* lib = nullptr; bool tried = false;
    decltype(&ncclGetUniqueId) getUniqueId = nullptr; decltype(&ncclCommInitRank) commInitRank = nullptr; decltype(&ncclCommDestroy) commDestroy = nullptr;
    decltype(&ncclAllGather) allGather = nullptr; decltype(&ncclSend) send = nullptr; decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclGroupStart) groupStart = nullptr; decltype(&ncclGroupEnd) groupEnd = nullptr; decltype(&ncclGetErrorString) errorString = nullptr;
};
static RcclApi* rcclApi() {
    static RcclApi api;
    if (api.tried) return api.lib ? &api : nullptr;
    api.tried = true;
    // the RCCL already in the process first (torch's own: a second copy from another path would be a second set of communicators' bookkeeping), then by soname
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { pdb::setError(std::string("pdb_comm: RCCL not found in this process (") + dlerror() + ")"); return nullptr; }
    bool ok = true;
    auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) ok = false; return p; };
    api.getUniqueId = (decltype(api.getUniqueId))sym("ncclGetUniqueId"); api.commInitRank = (decltype(api.commInitRank))sym("ncclCommInitRank");
    api.commDestroy = (decltype(api.commDestroy))sym("ncclCommDestroy"); api.allGather = (decltype(api.allGather))sym("ncclAllGather");
    api.send = (decltype(api.send))sym("ncclSend"); api.recv = (decltype(api.recv))sym("ncclRecv");
    api.groupStart = (decltype(api.groupStart))sym("ncclGroupStart"); api.groupEnd = (decltype(api.groupEnd))sym("ncclGroupEnd");
    api.errorString = (decltype(api.errorString))sym("ncclGetErrorString");
    if (!ok) { pdb::setError("pdb_comm: the RCCL in this process lacks an entry point"); return nullptr; }
    api.lib = h;
    return &api;
}
#define NCCLCHK(api, expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) { pdb::setError(std::string("RCCL: ") + (api)->errorString(_r) + " at " #expr); return PDB_ERR_HIP; } } while (0)
int pdb_comm_unique_id(void* id128) {
    if (!id128) { pdb::setError("pdb_comm_unique_id: null argument"); return PDB_ERR_ARG; }
    RcclApi* R = rcclApi(); if (!R) return PDB_ERR_NO_DEVICE;
    static_assert(sizeof(ncclUniqueId) == 128, "id size");
    ncclUniqueId id; NCCLCHK(R, R->getUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return PDB_OK;
}
static void commFree(pdb_batch* b) {
    RcclApi* R = rcclApi();
    for (int p = 0; p < PDB_MAX_PARTS; ++p) { if (b->comm[p] && R) (void)R->commDestroy(b->comm[p]); b->comm[p] = nullptr; }
    b->commWorld = 0;
}
int pdb_comm_init(pdb_batch* b, int world, int rank, const void* ids, int n_ids) {
    if (!b || world < 1 || rank < 0 || rank >= world || !ids || n_ids < b->parts || b->parts < 2) { pdb::setError("pdb_comm_init: bad argument (pdb_set_partitions first; one 128-byte id per partition, the same on every rank)"); return PDB_ERR_ARG; }
    RcclApi* R = rcclApi(); if (!R) return PDB_ERR_NO_DEVICE;
    commFree(b);
    HIPCHK(hipSetDevice(b->device));
    for (int p = 0; p < b->parts; ++p) {   // every rank creates the partitions' communicators in the same order
        ncclUniqueId id; memcpy(&id, (const uint8_t*)ids + (size_t)p * sizeof(id), sizeof(id));
        const ncclResult_t r = R->commInitRank(&b->comm[p], world, id, rank);
        if (r != ncclSuccess) {   // the communicators made so far go at once (the caller falls back on every rank: sharding.LibraryExchange agrees over the ranks)
            b->comm[p] = nullptr; commFree(b);
            pdb::setError(std::string("RCCL: ") + R->errorString(r) + " at ncclCommInitRank (partition " + std::to_string(p) + ")");
            return PDB_ERR_HIP;
        }
    }
    b->commWorld = world; b->commRank = rank;
    return PDB_OK;
}
int pdb_comm_destroy(pdb_batch* b) { if (!b) { pdb::setError("null argument"); return PDB_ERR_ARG; } commFree(b); return PDB_OK; }
// scatter_src (rank 0 only; device memory): float[world][cars of the partition][action stride], rank r's rows of this partition at index r.
// gathered (device memory, every rank): pdb_step_out[world][cars of the partition].  Both are read / written on the partition's stream.
int pdb_step_exchange_partition(pdb_batch* b, float dt, int part, const float* scatter_src, pdb_step_out* gathered) {
    if (!b || part < 0 || part >= b->parts || b->parts < 2 || !b->partStream[part] || !gathered) { pdb::setError("pdb_step_exchange_partition: bad argument"); return PDB_ERR_ARG; }
    if (!b->comm[part]) { pdb::setError("pdb_step_exchange_partition: pdb_comm_init first"); return PDB_ERR_ARG; }
    if (b->commRank == 0 && !scatter_src) { pdb::setError("pdb_step_exchange_partition: rank 0 passes the learner's action rows"); return PDB_ERR_ARG; }
    RcclApi* R = rcclApi(); if (!R) return PDB_ERR_NO_DEVICE;
    const int c0 = partFirst(b, part), c1 = partFirst(b, part + 1), c = c1 - c0;
    if (c <= 0) return PDB_OK;
    hipStream_t st = b->partStream[part];
    float* mine = b->dActions + (size_t)c0 * b->actionStride;
    const size_t rowFloats = (size_t)c * b->actionStride;
    if (b->commRank == 0) {   // the learner's rows for its own cars stay on the device; the other ranks' go out point to point
        HIPCHK(hipMemcpyAsync(mine, scatter_src, rowFloats * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (b->commWorld > 1) {
            NCCLCHK(R, R->groupStart());
            for (int r = 1; r < b->commWorld; ++r) NCCLCHK(R, R->send(scatter_src + (size_t)r * rowFloats, rowFloats, ncclFloat, r, b->comm[part], st));
            NCCLCHK(R, R->groupEnd());
        }
    } else NCCLCHK(R, R->recv(mine, rowFloats, ncclFloat, 0, b->comm[part], st));
    int rc = pdb_step_partition(b, dt, part, nullptr);
    if (rc != PDB_OK) return rc;
    static_assert(sizeof(pdb_step_out) % 4 == 0, "rows as floats");
    NCCLCHK(R, R->allGather(b->dOutActive + c0, gathered, (size_t)c * (sizeof(pdb_step_out) / 4), ncclFloat, b->comm[part], st));
    b->partEndStale[part] = true;   // the partition's end includes the gather (recorded by whoever waits: freshEnd)
    return PDB_OK;
}
int pdb_wait_partitions(pdb_batch* b, void* hip_stream) {
    if (!b) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : b->stream;
    if (s == b->stream) return joinParts(b);
    if (b->parts > 1)
        for (int p = 0; p < b->parts; ++p) if (b->partEnd[p] && partFirst(b, p + 1) > partFirst(b, p)) { if (int rcf = freshEnd(b, p)) return rcf; HIPCHK(hipStreamWaitEvent(s, b->partEnd[p], 0)); }
    return PDB_OK;
}
int pdb_partition_mark(pdb_batch* b) {
    if (!b) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    b->partMark = true;
    return PDB_OK;
}
int pdb_partition_elapsed_ms(pdb_batch* b, int part, float* ms, int* cars) {
    if (!b || !ms || part < 0 || part >= b->parts || b->parts < 2 || !b->partEnd[part]) { pdb::setError("pdb_partition_elapsed_ms: no such partition"); return PDB_ERR_ARG; }
    if (int rcf = freshEnd(b, part)) return rcf;
    HIPCHK(hipEventSynchronize(b->partEnd[part]));
    HIPCHK(hipEventElapsedTime(ms, b->partStart[part], b->partEnd[part]));
    if (cars) *cars = partFirst(b, part + 1) - partFirst(b, part);
    return PDB_OK;
}

int pdb_event_record(pdb_batch* b, int which) {
    if (!b) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    if (int rcj = joinParts(b)) return rcj;
    HIPCHK(hipEventRecord(which ? b->tev1 : b->tev0, b->stream));
    return PDB_OK;
}

int pdb_event_elapsed_ms(pdb_batch* b, float* ms) {
    if (!b || !ms) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    HIPCHK(hipEventSynchronize(b->tev1));
    HIPCHK(hipEventElapsedTime(ms, b->tev0, b->tev1));
    return PDB_OK;
}

int pdb_sync(pdb_batch* b) {
    if (!b) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    if (int rcj = joinParts(b)) return rcj;
    HIPCHK(hipStreamSynchronize(b->stream));
    return PDB_OK;
}

int pdb_step_host(pdb_batch* b, const float* actions, float dt, pdb_step_out* out) {
    if (!b || !actions) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    if (int rcj = joinParts(b)) return rcj;   // partition kernels still in flight read the action block
    HIPCHK(hipMemcpyAsync(b->dActions, actions, sizeof(float) * b->actionStride * (size_t)b->n, hipMemcpyHostToDevice, b->stream));
    int rc = launch(b, dt, true);
    if (rc != PDB_OK) return rc;
    if (out) HIPCHK(hipMemcpyAsync(out, b->dOutActive, sizeof(pdb_step_out) * (size_t)b->n, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return PDB_OK;
}

// One tick of the cars whose hold byte is zero; the others sit the launch out (record, contact joints, output row untouched).  The vector env's
// same-step reset (projectd_sb3.py): the lanes whose episode ended in the step just taken run their reset tick -- teleport + step([0, 0]),
// projectd_env.py:216-227 -- before the step returns, while every other lane is held.  Workgroups whose cars are all held leave at once, so the
// launch costs what the few cars that move cost.
int pdb_step_host_held(pdb_batch* b, const float* actions, float dt, const uint8_t* hold, pdb_step_out* out) {
    if (!b || !actions || !hold) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    if (int rcj = joinParts(b)) return rcj;
    if (!b->dHold) HIPCHK(hipMalloc(&b->dHold, (size_t)b->n));
    HIPCHK(hipMemcpyAsync(b->dHold, hold, (size_t)b->n, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipMemcpyAsync(b->dActions, actions, sizeof(float) * b->actionStride * (size_t)b->n, hipMemcpyHostToDevice, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));   // nothing in flight reads the constants block while the mask is switched on ...
    b->K.holdMask = b->dHold;
    if (int rck = pushK(b, b->stream, false)) return rck;
    int rc = launch(b, dt, true);
    if (rc == PDB_OK && out && hipMemcpyAsync(out, b->dOutActive, sizeof(pdb_step_out) * (size_t)b->n, hipMemcpyDeviceToHost, b->stream) != hipSuccess) { pdb::setError("pdb_step_host_held: output copy failed"); rc = PDB_ERR_HIP; }
    if (hipStreamSynchronize(b->stream) != hipSuccess && rc == PDB_OK) { pdb::setError("pdb_step_host_held: the held step failed on the device"); rc = PDB_ERR_HIP; }
    b->K.holdMask = nullptr;                   // ... and off again, whatever happened in between
    if (int rck = pushK(b, b->stream, false)) return rck;
    return rc;
}

int pdb_get_car_state(pdb_batch* b, int first, int count, pdb_car_state* out) {
    if (!b || !out || first < 0 || count < 0 || first + count > b->n) { pdb::setError("bad range"); return PDB_ERR_ARG; }
    if (int rcj = joinParts(b)) return rcj;
    HIPCHK(hipMemcpyAsync(out, b->dCarStates + first, sizeof(pdb_car_state) * (size_t)count, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    return PDB_OK;
}

#ifdef PDB_STAMPS
int pdb_debug_stamps(pdb_batch* b, unsigned long long* out) {
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipMemcpy(out, b->dStamps, sizeof(unsigned long long) * 64 * (size_t)b->n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(b->dStamps, 0, sizeof(unsigned long long) * 64 * (size_t)b->n));
    return PDB_OK;
}
#endif

// Measurement: HIP events around every `every`-th first-pass launch of each launch site (partition streams and the batch's own), recorded on the stream the
// kernel is launched on; at most PDB_KERNEL_SAMPLES per site between two reads.  0 switches it off.  Launches recorded into a graph are not sampled.
int pdb_sample_kernel(pdb_batch* b, int every) {
    if (!b || every < 0) { pdb::setError("pdb_sample_kernel: bad argument"); return PDB_ERR_ARG; }
    HIPCHK(hipSetDevice(b->device));
    if (every > 0)
        for (int q = 0; q <= PDB_MAX_PARTS; ++q) for (hipEvent_t& e : b->samples[q].ev) if (!e) HIPCHK(hipEventCreate(&e));
    for (int q = 0; q <= PDB_MAX_PARTS; ++q) { b->samples[q].n = 0; b->samples[q].tick = 0; }
    b->sampleEvery = every;
    return PDB_OK;
}
// waits for the sampled launches; average duration in microseconds, their number and the average number of cars per sampled launch; clears the samples
int pdb_sampled_kernel_us(pdb_batch* b, double* avg_us, int* samples, double* avg_cars) {
    if (!b) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    double sum = 0, cars = 0; int cnt = 0;
    for (int q = 0; q <= PDB_MAX_PARTS; ++q) {
        KernelSamples& KS = b->samples[q];
        for (int i = 0; i < KS.n; ++i) {
            HIPCHK(hipEventSynchronize(KS.ev[2 * i + 1]));
            float ms = 0; HIPCHK(hipEventElapsedTime(&ms, KS.ev[2 * i], KS.ev[2 * i + 1]));
            sum += ms; cars += KS.cars[i]; ++cnt;
        }
        KS.n = 0;
    }
    if (avg_us) *avg_us = cnt ? sum * 1000.0 / cnt : 0.0;
    if (samples) *samples = cnt;
    if (avg_cars) *avg_cars = cnt ? cars / cnt : 0.0;
    return PDB_OK;
}

int pdb_kernel_time_us(pdb_batch* b, double* avg_us, int* launches) {
    if (!b) { pdb::setError("null argument"); return PDB_ERR_ARG; }
    if (avg_us) *avg_us = b->kernelLaunches ? (b->kernelMs * 1000.0 / b->kernelLaunches) : 0.0;
    if (launches) *launches = b->kernelLaunches;
    b->kernelMs = 0; b->kernelLaunches = 0;
    return PDB_OK;
}

}  // extern "C"
