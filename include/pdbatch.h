/* pdbatch C ABI -- the drop-in boundary of the MI355X-native batched vehicle stepper.
 *
 * One pdb_batch = N independent (simulator, car) pairs of the reference, all on one track with one
 * car model, resident on one GPU.  Each entry point names the reference interface it replaces
 * (reference src/PyProjectD/PyProjectD.cpp unless stated otherwise):
 *
 *   pdb_build_car_model      addCar(simId, model)              :219-237 -> Car::init        (load-time half)
 *   pdb_set_car_tune         setCarTune / setCarRawTune        :328-345 -> SetupManager::setTune/setRawTune
 *   pdb_set_scoring_var      setScoringVar / getScoringVar     :347-365
 *   pdb_build_track          loadTrack(simId, name)            :186-203 -> Track::init
 *   pdb_initial_state        addCar + teleportCarByMode(Start) :219-237,283-290
 *   pdb_teleport_to_spline   teleportCarToSpline               :274-281
 *   pdb_teleport_to_location teleportCarToLocation             :250-257 -> Car::forcePosition (Car.cpp:1240-1272)
 *   pdb_teleport_to_pit      teleportCarToPits                 :259-266 -> Car::teleportToPits (Car.cpp:1310-1323; pits.ini: Sim/Track.cpp:151-175)
 *   pdb_teleport_by_mode     teleportCarByMode(simId, carId, mode)  :283-290 -> Car::teleportByMode (Start / Nearest / Random)
 *   pdb_set_auto_teleport    setCarAutoTeleport                :292-295 (teleport inside the tick, ScoringSystem.cpp:194-225)
 *   pdb_create / pdb_destroy createSimulator / destroySimulator:111-149 (x N)
 *   pdb_set_assists          setCarAssists                     :307-317
 *   pdb_step / pdb_step_host setCarControls + stepSimulator + getCarState   :297-305,160-180,319-326
 *   pdb_get_car_state        getCarState                       :319-326 (664-byte CarState per car)
 *   pdb_reset / pdb_reset_mode / pdb_reset_device / pdb_reset_mask_device
 *                            teleportCarByMode for a mask of cars (projectd_env.py:216-227), on the device
 *   pdb_set_seed             setSeed                           :50-53
 *   pdb_step_ring / pdb_step_partition / pdb_set_partitions   no counterpart: N stepSimulator loops of N independent
 *                            simulators, which the reference runs as N processes, here as free-running car ranges
 *   body contacts            PhysicsEngineODE::collisionStep + Car::onCollisionCallback run inside every step entry point
 *                            (pdb_car_params.collider, loaded by pdb_build_car_model from colliders.ini / collider.bin)
 *
 * Conventions: plain pointers and sizes, caller-owned buffers, no exceptions across the ABI.
 * Functions returning int give 0 on success and a negative pdb_status on failure;
 * pdb_last_error() returns the message of the calling thread's last failure.
 * There is NO CPU fallback: device entry points fail with PDB_ERR_NO_DEVICE when no GPU is usable.
 */
#ifndef PDBATCH_H
#define PDBATCH_H
#include "pdb_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pdb_batch pdb_batch;

enum pdb_status {
    PDB_OK = 0,
    PDB_ERR_ARG = -1,        /* bad argument */
    PDB_ERR_IO = -2,         /* missing / malformed content file, unsupported car feature */
    PDB_ERR_NO_DEVICE = -3,  /* no usable HIP device */
    PDB_ERR_HIP = -4         /* HIP runtime error */
};

enum pdb_action_mode {
    PDB_ACTION_CONTROLS = 0, /* actions[i] = {steer, gas}          (CarControls fields) */
    PDB_ACTION_ENV = 1,      /* actions[i] = {a0, a1}: steer = a0, gas = linscale(a1,-1,1,0.1,1) (projectd_env.py:159-160) */
    PDB_ACTION_FULL = 2      /* actions[i] = 8 floats: steer, clutch, brake, handBrake, gas, requestedGearIndex (-1 = none),
                              * gearUp, gearDn -- every CarControls field of setCarControls (PyProjectD.cpp:297-305) */
};

const char* pdb_last_error(void);
const char* pdb_version(void);

/* ---- host-side content (no device needed) ---- */
int pdb_build_car_model(const char* base_path, const char* model_name, pdb_car_params* out);
int pdb_set_car_tune(pdb_car_params* params, const char* base_path, const char* model_name, const char* name, float value, int raw);
int pdb_set_scoring_var(pdb_car_params* params, const char* name, float value);
float pdb_get_scoring_var(const pdb_car_params* params, const char* name);
int pdb_set_assists(pdb_car_params* params, int auto_clutch, int auto_shift, int auto_blip, int smooth_steer);
/* *blob is malloc'ed; release with pdb_free */
/* diagnostic: the step's reproducible elementary functions evaluated on the host, fn = 0 sin, 1 cos, 2 tan, 3 atan,
 * 4 atan2(x,y), 5 asin, 6 acos, 7 pow(x,y); y may be NULL for the one-argument functions */
int pdb_math_eval(int fn, const float* x, const float* y, float* out, int n);
int pdb_build_track(const char* base_path, const char* track_name, void** blob, uint64_t* bytes);
/* Track::init with options.  PDB_TRACK_RECOMPUTE_FAT_POINTS ignores spline.cache and runs Track::computeFatPoints
 * (Sim/Track.cpp:366-467, incl. the ray-traced sides of TRACE_SIDES=1 tracks) the way the reference does when the cache
 * is missing or stale (Track.cpp:209-218); nothing is written back to the content directory. */
#define PDB_TRACK_RECOMPUTE_FAT_POINTS 1
int pdb_build_track_opts(const char* base_path, const char* track_name, int flags, void** blob, uint64_t* bytes);
void pdb_free(void* p);
int pdb_initial_state(const pdb_car_params* params, const void* track_blob, pdb_dyn_state* out);
int pdb_teleport_to_spline(const pdb_car_params* params, const void* track_blob, float distance_norm, pdb_dyn_state* inout);
/* teleportCarToLocation = Car::forcePosition((x, y, z)): the car keeps its orientation, is put down on what a ray from 10 m above the point hits
 * (at the point's own height if it hits nothing) and is reset like by every other teleport */
int pdb_teleport_to_location(const pdb_car_params* params, const void* track_blob, float x, float y, float z, pdb_dyn_state* inout);
/* teleportCarToPits = Car::teleportToPits(pit_id): heading and position of pit box `pit_id` of the track's pits.ini; an id outside the list leaves
 * the car alone, like the reference.  pdb_track_num_pits: how many boxes the blob carries; pdb_track_pit: a box's matrix (the reference's mat44f,
 * 16 floats row by row) */
int pdb_teleport_to_pit(const pdb_car_params* params, const void* track_blob, int pit_id, pdb_dyn_state* inout);
int pdb_track_num_pits(const void* track_blob);
int pdb_track_pit(const void* track_blob, int pit_id, float* m16);
/* Car::teleportByMode: mode 0 = Start, 1 = Nearest (the car's trackLocation), 2 = Random (the car's own rand() state,
 * pdb_dyn_state.randState: the C runtime's generator of the reference build, seeded 1 like an unseeded process) */
int pdb_teleport_by_mode(const pdb_car_params* params, const void* track_blob, int mode, pdb_dyn_state* inout);
/* setCarAutoTeleport: teleport by `mode` inside the tick that raises the collision flag / finds the car off the track */
int pdb_set_auto_teleport(pdb_car_params* params, int on_collision, int on_bad_location, int mode);

/* ---- device batch ---- */
pdb_batch* pdb_create(int device, int n_cars, const pdb_car_params* params, const void* track_blob, uint64_t track_bytes,
                      int action_mode);
void pdb_destroy(pdb_batch* b);
int pdb_num_cars(const pdb_batch* b);
/* every car <- *state (broadcast) or states[first..first+count) */
int pdb_set_state_all(pdb_batch* b, const pdb_dyn_state* state);
int pdb_set_state(pdb_batch* b, int first, int count, const pdb_dyn_state* states);
int pdb_get_state(pdb_batch* b, int first, int count, pdb_dyn_state* states);
/* the cars' live contact joints (PhysicsEngineODE's contactGroupDynamic): PDB_MAX_CONTACTS pdb_contact per car, the first
 * pdb_dyn_state.numContacts of a car's row are alive.  A snapshot that must replay bit for bit across an odd -> even frame
 * boundary carries them next to the state records. */
int pdb_get_contacts(pdb_batch* b, int first, int count, pdb_contact* out);
int pdb_set_contacts(pdb_batch* b, int first, int count, const pdb_contact* in);
/* teleportCarByMode(Start) for cars with mask[i] != 0 (mask == NULL: all cars); host mask.  The teleport itself runs on the
 * device (only the mask crosses PCIe); returns when it is done. */
int pdb_reset(pdb_batch* b, const uint8_t* mask);
/* the same for any Car::teleportByMode mode: 0 Start, 1 Nearest (each car's trackLocation), 2 Random (each car's rand() state) */
int pdb_reset_mode(pdb_batch* b, const uint8_t* mask, int mode);
/* the same with the mask already on the device (e.g. a learner's `terminated` tensor): asynchronous on pdb_stream, no host
 * round trip */
int pdb_reset_device(pdb_batch* b, const uint8_t* device_mask, int mode);
/* Deferred resets without any extra launch: a device array of n_cars bytes owned by the batch; a car whose byte is 1 + mode is
 * teleported at the top of its next tick (which then steps it: with the env's zero action this is projectd_env.py:216-227
 * `reset()` = teleport + step([0,0])), and that tick clears the byte.  Asking for the pointer arms the check in the step
 * kernels. */
uint8_t* pdb_reset_mask_device(pdb_batch* b);
/* env mode: clear the episode sums (cumulative reward, step count, pending reset) of the cars with device_mask[i] != 0 (NULL: all),
 * on the device, asynchronous on pdb_stream -- the bookkeeping half of the env's reset() (projectd_env.py:216-227) */
int pdb_clear_episodes(pdb_batch* b, const uint8_t* device_mask);
/* seconds without a new track point before pdb_step_out.flags bit 2 (stuck) rises: projectd_env.py stuck_timeout, default 5 */
int pdb_set_stuck_timeout(pdb_batch* b, double seconds);
/* Env mode: the reward / termination / reset bookkeeping of pyprojectd/projectd_env.py:173-227 per car inside the tick (see
 * pdb_env_config): pdb_step_out.reward is the env's reward, flags bit 3 = terminated, bit 4 = reset tick; a terminated car's next
 * tick teleports it (teleport_on_reset) and steps it with the zero action.  ProjectDVecEnv.step() = one launch. */
int pdb_set_env(pdb_batch* b, const pdb_env_config* cfg);
/* setSeed (PyProjectD.cpp:50-53 = srand) per car: the state of the car's own C-runtime rand(), used by the Random teleport mode */
int pdb_set_seed(pdb_batch* b, const uint32_t* seeds);
/* device pointers owned by the batch: float actions[N][2], pdb_step_out out[N] */
float* pdb_actions_device(pdb_batch* b);
pdb_step_out* pdb_out_device(pdb_batch* b);
/* redirect the per-tick outputs into a caller-owned device block of n_cars pdb_step_out (e.g. one slot of a trajectory ring
 * that is gathered to the learner every k ticks); NULL restores the library's own block */
int pdb_set_out_device(pdb_batch* b, pdb_step_out* out);
/* Free-running partitions.  Cars are independent (one car per simulator in the reference): nothing orders one car's tick
 * against another's.  pdb_set_partitions cuts the batch into 1..4 contiguous parts, each with its own HIP stream;
 * pdb_step_ring enqueues n_ticks ticks of every car, part by part, and the parts' kernels run concurrently and drift against
 * each other -- one part's ramp-up / drain and inter-kernel gap is filled by the other's work.  Ordering: every part starts
 * after the work enqueued on the batch's stream so far; with join != 0 the batch's stream continues only after all parts have
 * done their n_ticks, with join == 0 it is not held back (the parts keep drifting across calls) and whoever consumes the
 * results orders itself with pdb_wait_partitions(b, stream): that stream then waits for every part's last enqueued kernel
 * (stream == NULL: the batch's stream).  The library's own entry points that work through the batch's stream (state reads and
 * writes, resets, plain steps, pdb_sync) order themselves after the parts' kernels in flight.  Tick i writes its outputs to ring + ((first_slot + i) % ring_slots) * n_cars; ring == NULL: the active
 * output block, every tick.  With one part this is n_ticks plain launches on the batch's stream.  Results do not depend on
 * the partitioning.  pdb_partition_mark / pdb_partition_elapsed_ms: HIP-event time of one part's kernels between the mark
 * and the last pdb_step_ring (synchronises on that part), and the number of cars in the part.
 * Streams: a process has four hardware queues and its null stream holds one; kernels of two streams that share a queue run one after the other.
 * While the batch runs on the library's own stream (no pdb_set_stream), part 0 runs on that stream, so that three parts are three streams of the
 * library's; after pdb_set_stream every part has a stream of its own (the caller's stream + three parts: keep other streams of the process idle
 * while they step, or use two parts -- DESIGN.md section 7).
 * ALIASING: while part 0 runs on the batch's own stream, pdb_partition_stream(b, 0) == pdb_stream(b): pdb_step_ring with join == 0 and
 * pdb_step_partition(b, dt, 0, ...) then DO queue behind (and hold back) whatever else is put on the batch's stream, and a caller's work on that stream
 * serialises with part 0.  pdb_set_stream gives part 0 a NEW stream: query pdb_partition_stream again after it -- a handle taken before (a torch
 * ExternalStream, a communicator or graph keyed to it) is stale. */
int pdb_set_partitions(pdb_batch* b, int parts);
/* diagnostic: how many cars the most recent contact pass of a launch site held (site = partition index, 4 = the batch's own
 * stream); read without waiting for anything, so it lags the launches still in flight */
int pdb_contact_pass_load(pdb_batch* b, int site);
/* The contact pass's grid (workgroups per launch): 0 = follow the load (the default: the host sizes every launch from the number of cars the last passes
 * held); > 0 = fixed -- for a caller that records the launches of pdb_step_partition into a graph of its own (a captured grid cannot follow the load; a
 * pass with fewer workgroups than queued car groups takes them in turn) */
int pdb_set_contact_grid(pdb_batch* b, int workgroups);
/* A car block of its own for one partition (NULL: back to the batch's): same car model and rigid-body topology, different tunes,
 * assists, scoring weights, auto-teleport -- what the reference gives every simulator separately (PyProjectD.cpp:328-365) --
 * e.g. to randomise the setup over the cars of a batch.  The partition's cars step, reset and teleport with it, through every
 * step entry point (the whole-batch ones then issue one launch per partition range). */
int pdb_set_partition_params(pdb_batch* b, int part, const pdb_car_params* params);
/* Multi-car simulators (Sim/Simulator.cpp:59-60,112-153: 1..100 cars share one Simulator; PyProjectD.cpp:219-237 addCar once per car): the batch's cars form worlds of
 * cars_per_world consecutive lanes (a divisor of the batch's car count; partitions then cut on whole worlds).  What couples the cars of a world on this path is the
 * slipstream -- Car::updateAirPressure (Car.cpp:557-585): the air density a car's wings and body meet is thinned by the wakes the OTHER cars of its simulator left at
 * the end of the last tick (Sim/SlipStream.cpp).  Body contacts between cars (PhysicsEngineODE.cpp:236-241) are not built.  pdb_get_slipstreams / pdb_set_slipstreams:
 * the wakes as state, both buffers ([0][count] then [1][count]; a car's next tick reads the one of its pdb_dyn_state.simFrame's parity) -- a new car's are zero. */
int pdb_set_world_size(pdb_batch* b, int cars_per_world);
int pdb_get_slipstreams(pdb_batch* b, int first, int count, pdb_slip_state* out);
int pdb_set_slipstreams(pdb_batch* b, int first, int count, const pdb_slip_state* in);
/* The device law: the NEXT tick's two action components (steer, throttle: the action modes with two floats a car) worked out by the tick's own launches, where they
 * write the car's observation row:   a[c] = bias[c] + sum_{k < 24} obs[k] * weights[k][c]   (the 24 products as the leaves of a balanced binary tree over 32 slots
 * in slot order; IEEE single, no fused operations) -- scripted input streams (BASELINE configs[2]: bias = row pdb_dyn_state.lawTick of `table`, [period][n_cars][2]
 * floats, the car's row counter advancing by one a tick and wrapping at `period`) and linear feedback laws of the observation (table == NULL: bias = bias0, or zero)
 * then cost a stepping stream no launch between two ticks.  The reference has no counterpart: there the actions come from the caller through
 * setCarControls (PyProjectD.cpp:297-305) every tick, and they still can -- the law only WRITES the batch's action buffer (pdb_actions_device), after the tick has read it;
 * the env mode's reset tick still steps with the zero action (projectd_env.py:222).  weights: [24][2] floats on the host (NULL: no law, the default); bias0: 2 floats
 * on the host or NULL; table: on the host (copied) or, with table_on_device != 0, device memory the caller keeps alive and may rewrite between ticks. */
int pdb_set_law(pdb_batch* b, const float* weights, const float* bias0, const float* table, int period, int table_on_device);
/* Per-LANE setup and reward weights (pdb_lane_tune, include/pdb_types.h): the reference's setCarTune / setScoringVar are per simulator, i.e. per env
 * (PyProjectD.cpp:328-365); here a lane's row overrides the eight tunes of projectd_env.py:127-130 and the scoring variables of its car block.
 * pdb_lane_tune_from_params (host library too) reads the row out of a block that went through pdb_set_car_tune / pdb_set_scoring_var;
 * pdb_set_lane_tunes(b, first, count, rows) installs rows for the cars [first, first + count) (rows == NULL: those lanes back to their block). */
int pdb_lane_tune_from_params(const pdb_car_params* params, pdb_lane_tune* row);
int pdb_set_lane_tunes(pdb_batch* b, int first, int count, const pdb_lane_tune* rows);
/* The rest of the setup, lane by lane (pdb_lane_setup, include/pdb_types.h: brake power, differential preload, gear ratios, anti-roll bars, rev limiter, turbo settings,
 * dampers, springs, bump stops, rod lengths, packers, toe and camber per wheel -- with pdb_lane_tune every SetupManager tune, Car/SetupManager.cpp:10-120).
 * pdb_lane_setup_from_params (host library too) reads the row out of a block that went through pdb_set_car_tune.  The first pdb_set_lane_setups of a batch builds the
 * table -- every lane's row from its own block (the batch's, or its partition's) -- and switches the batch to the kernel pair compiled for it; rows == NULL puts the
 * lanes [first, first + count) back to their blocks' values.  (Cars of the 40-row kernel class -- DynamicController files, brake temperatures, more than 33
 * constraint rows -- keep their two kernel pairs, which test for the table at run time; a controller that drives a tunable value overrides the row, as it overrides
 * the block.)  pdb_set_partition_params is refused once the table exists (install the partitions' blocks first). */
int pdb_lane_setup_from_params(const pdb_car_params* params, pdb_lane_setup* row);
int pdb_set_lane_setups(pdb_batch* b, int first, int count, const pdb_lane_setup* rows);
/* join: bit 0 = the batch's stream waits for the ring's kernels; bit 1 = the partitions do NOT wait for what is queued on the batch's stream (the caller orders
 * its own dependencies on pdb_partition_stream; asynchronous resets queued through the library are still waited for) */
int pdb_step_ring(pdb_batch* b, float dt, int n_ticks, pdb_step_out* ring, int ring_slots, int first_slot, int join);
int pdb_wait_partitions(pdb_batch* b, void* hip_stream);
/* Per-partition loops: pdb_step_partition enqueues one tick of one part on that part's stream (pdb_partition_stream), nothing
 * forked or joined; a caller that also puts its own per-tick work for the part's cars (pdb_partition_range) on that stream --
 * a policy evaluated from the part's observation rows -- keeps each part's whole closed loop independent of the others.
 * Ordering: the part's stream is ordered after a pdb_reset_device queued on the batch's stream; any other asynchronous work the
 * caller puts on the batch's stream (pdb_set_stream) and wants done before the part's tick, the caller orders itself (an event
 * on pdb_partition_stream). */
int pdb_step_partition(pdb_batch* b, float dt, int part, pdb_step_out* out);
void* pdb_partition_stream(pdb_batch* b, int part);
/* The learner exchange of a sharded job (one process per GPU, contiguous car blocks; no counterpart in the reference, whose envs are
 * separate processes under stable-baselines' SubprocVecEnv) issued by the library itself: every partition has its own RCCL communicator, and
 * pdb_step_exchange_partition enqueues on the partition's stream
 *     rank 0's action rows for this partition -> every rank's action block  |  one tick of the partition  |  all-gather of its output rows,
 * three enqueues from C, nothing of the host's in between, the partitions never joined.  RCCL is taken from the process at run time (the one
 * torch uses, if torch is loaded); without it these return PDB_ERR_NO_DEVICE and nothing else is affected.
 *   pdb_comm_unique_id: rank 0 calls it once per partition and hands the 128-byte ids to every rank (any transport: a torch.distributed
 *     broadcast, a file); pdb_comm_init(b, world, rank, ids, n_ids >= partitions) is collective over the ranks, after pdb_set_partitions.
 *   scatter_src (rank 0 only, device memory): float[world][cars of the partition][action stride] -- rank r's rows at index r;
 *   gathered (every rank, device memory): pdb_step_out[world][cars of the partition], valid once the partition's stream has passed the call. */
int pdb_comm_unique_id(void* id128);
int pdb_comm_init(pdb_batch* b, int world, int rank, const void* ids, int n_ids);
int pdb_comm_destroy(pdb_batch* b);
int pdb_step_exchange_partition(pdb_batch* b, float dt, int part, const float* scatter_src, pdb_step_out* gathered);
/* A policy that lives on the HOST (BASELINE configs[4]; the reference's loop: projectd_env.py:157-171 under stable-baselines),
 * pipelined over the partitions instead of one synchronous round trip per tick (pdb_step_host): the library owns page-locked
 * host mirrors of the action block (pdb_host_actions: float[N][stride]) and of the output block (pdb_host_out: pdb_step_out[N]).
 * pdb_step_host_partition enqueues, on the partition's stream, the upload of the partition's action rows from the mirror, one
 * tick of its cars and the download of its output rows into the mirror, and returns at once; pdb_wait_host_partition blocks
 * until that download has landed.  The host then computes partition p's next actions from its rows while the other
 * partitions' ticks and copies are in flight: no device-wide synchronisation per tick.  The trajectories are those of
 * pdb_step_host with the same actions (cars are independent). */
float* pdb_host_actions(pdb_batch* b);
pdb_step_out* pdb_host_out(pdb_batch* b);
int pdb_step_host_partition(pdb_batch* b, float dt, int part);
int pdb_wait_host_partition(pdb_batch* b, int part);
int pdb_partition_range(pdb_batch* b, int part, int* first, int* count);
int pdb_partition_mark(pdb_batch* b);
int pdb_partition_elapsed_ms(pdb_batch* b, int part, float* ms, int* cars);
void* pdb_stream(pdb_batch* b);
/* one tick of every car, reading pdb_actions_device and writing pdb_out_device; asynchronous on pdb_stream */
int pdb_step(pdb_batch* b, float dt);
/* n back-to-back ticks with the current actions (launch-overhead-free replay of a captured graph when n > 1) */
int pdb_step_n(pdb_batch* b, float dt, int n);
/* same as pdb_step but returns right after the launch (no host wait) */
int pdb_step_async(pdb_batch* b, float dt);
/* run the batch on a caller-owned HIP stream (e.g. torch's current stream) instead of its own */
int pdb_set_stream(pdb_batch* b, void* hip_stream);
/* HIP events on the batch stream bracketing a timed region: which = 0 (begin) / 1 (end) */
int pdb_event_record(pdb_batch* b, int which);
/* waits for the end event and returns the elapsed milliseconds between the two events */
int pdb_event_elapsed_ms(pdb_batch* b, float* ms);
int pdb_sync(pdb_batch* b);
/* pdb_step_host for a SUBSET of the cars: those whose hold byte (hold[n], host memory) is non-zero sit the tick out -- record, contact joints and
 * output row untouched; `out` receives every row (the held cars' rows as they were).  No counterpart in the reference, where every env is its own
 * simulator and resets on its own (projectd_env.py:216-227); a vector env uses it to run the reset tick of the lanes whose episode has just ended
 * before its step returns (stable-baselines3's convention), while every other lane waits. */
int pdb_step_host_held(pdb_batch* b, const float* actions, float dt, const uint8_t* hold, pdb_step_out* out);
/* convenience: upload actions, one tick, download outputs */
int pdb_step_host(pdb_batch* b, const float* actions, float dt, pdb_step_out* out);
int pdb_get_car_state(pdb_batch* b, int first, int count, pdb_car_state* out);
/* Measurement (bench.py's roofline block): HIP events around every `every`-th first-pass launch (pdb_step_kernel*) of each launch site -- the
 * partitions' streams and the batch's own -- recorded on the stream the kernel is launched on; 0 = off; at most 64 samples per site between two reads.
 * pdb_sampled_kernel_us waits for the sampled launches and returns their average duration (microseconds), their number and the average number of
 * cars per sampled launch, and clears the samples. */
int pdb_sample_kernel(pdb_batch* b, int every);
int pdb_sampled_kernel_us(pdb_batch* b, double* avg_us, int* samples, double* avg_cars);
/* average device time of the step kernel over the launches since the last call, in microseconds (HIP events) */
int pdb_kernel_time_us(pdb_batch* b, double* avg_us, int* launches);

#ifdef __cplusplus
}
#endif
#endif
