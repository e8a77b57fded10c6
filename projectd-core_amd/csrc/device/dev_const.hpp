// pdbatch device side: constants block shared by the host launcher and every kernel instantiation.
#pragma once
#include "dev_math.hpp"
#include "pmath.hpp"

#define PDB_WAVE 64
#define PDB_CH_SLOTS 60          // chassis contribution slots in the reference's call order: 4x5 suspension, 4 tyre, 2x8 heave, 6x2 wings, 1 axle reaction, 4 ARB (+ spare)
#define PDB_HUB_SLOTS 12         // per hub: 4 suspension, 3 tyre, 4 heave, 1 ARB
#define PDB_AXLE_SLOTS 20

struct DevConst {
    float camC[4], camS[4], camM33[4];   // cos/sin of the static camber and ((1-c)+c) (mat44f::createFromAxisAngle about z)
    float acos096;                        // pm::acosf_(0.96f)
    int rowStart[PDB_MAX_JOINTS + 1];
    int rowFirst[PDB_MAX_ROWS];                     // 1: the row is its joint's first
    int rowB0[PDB_MAX_ROWS], rowB1[PDB_MAX_ROWS];   // bodies of each constraint row (static per model: scalar loads in the A assembly)
    // per body and row: byte offset of the body's six Jacobian columns inside the row (0 = first body of the row, 24 = second), 255 = the
    // row does not touch the body.  48 bytes per body: three 16-byte loads per lane in the constraint-force walk (lane = body x component)
    __attribute__((aligned(16))) unsigned char cfOff[PDB_MAX_BODIES][48];
    float dt;
    float fps;   // 1.0f / dt
    float invMass[PDB_MAX_BODIES], invInertia[PDB_MAX_BODIES][3];   // 1.0f / mass, 1.0f / inertia: divided once on the host (IEEE single division on both sides)
    double dtD;
    double stuckTimeout;   // seconds without a new track point before pdb_step_out.flags bit 2 rises (projectd_env.py stuck_timeout = 5.0)
    // env mode (pdb_set_env): projectd_env.py:173-227 evaluated per car inside the tick
    double envHitPenalty, envOffPenalty, envStuckPenalty, envLowReward;
    int envMode, envTermHit, envTermOff, envTermStuck, envTeleportOnReset, envTeleportMode;
    int actionMode;
    int wantCarState;
    pdb_lane_tune laneDefault;            // the car block's own values of the per-lane tunes (a lane without a valid row reads these)
    const pdb_lane_tune* laneTunes;       // [cars of the batch] or null: pdb_set_lane_tunes
    const pdb_lane_setup* laneSetups;     // [cars of the batch] or null: pdb_set_lane_setups (read by the kernel pair compiled for it only)
    const unsigned char* holdMask;        // [cars of the batch] or null: cars whose byte is non-zero sit this launch out (pdb_step_host_held: the reset tick of the lanes whose episode just ended)
    const pdb_dyn_state* freshState;      // the record of a fresh car at the start pose (device memory): env mode re-creates a car whose pose is no longer finite from it
    // multi-car simulators (pdb_set_world_size): a world = worldSize consecutive cars that share one Simulator in the reference; their only coupling on this path is the
    // slipstream (Car::updateAirPressure).  slip: [2][slipStride] -- a tick reads the buffer of its frame's parity (what the cars' last postStep left) and writes the other
    int worldSize;
    int slipStride;
    pdb_slip_state* slip;
    unsigned long long* stamps;   // diagnostic build only (-DPDB_STAMPS): [car][32] shader-clock stamps of the first pass, then [stampCars + car][32] of the contact pass
    int stampCars;
    // device law (pdb_set_law): the NEXT tick's two action components evaluated where the tick writes its observation row out -- a[c] = bias[c] + sum_k obs[k] * lawW[k][c],
    // the 24 products summed as the leaves of waveSumF's balanced tree in slot order; bias = lawBias0, or row (record.lawTick) of the table lawBias ([lawPeriod][lawStride cars][2],
    // offset to the launch's first car).  A scripted input stream or a linear feedback law then costs the stream no launch of its own.  lawPeriod == 0: no law (the caller writes the actions)
    int lawPeriod;
    int lawStride;
    const float* lawBias;
    float lawBias0[2];
    float lawW[2 * 24];
    int noTeam;   // diagnostic (PDB_NO_TEAM in the environment at pdb_create): the car waves each walk their own car's joint rows, bars, wings ... as before round 5 (the form a model with more than 21 joints takes)
};

