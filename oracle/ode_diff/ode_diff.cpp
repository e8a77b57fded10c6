// ORACLE / TEST INFRASTRUCTURE -- differential test of oracle/rb (pdrb, this project's restatement of ODE's stepper) against a REAL
// Open Dynamics Engine 0.16.x, for whoever has one.  ODE is not in the reference tree and not in this image, so
// this program cannot be linked here: it is committed so that the "recalled" semantics of the Fixed / Ball / Slider / DBall rows,
// the contact rows and the finite-rotation integrator are FALSIFIABLE -- build it where libode lives (see Makefile), run it, and
// every number it prints is a direct pdrb-vs-ODE comparison.  tests/test_ode_diff.py runs it when the binary exists, and checks
// here (against the ODE headers shipped under the reference's thirdparty/, when present) that it at least compiles.
//
// What it does: loads a car's constant block (projectd-core_amd/data/*.pdcar = pdb_car_params) and an initial state record
// (pdb_dyn_state, written by make_inputs.py), builds the same bodies and joints in ODE (through the very calls the reference makes:
// Physics/ODE/RigidBodyODE.cpp:9-98, JointODE.cpp:21-89, PhysicsEngineODE.cpp:23-29) and in pdrb, then steps both for N ticks of
// 1/333 s under identical scripted external forces (gravity, a spring holding every hub / axle off the ground, a rocking force on
// the chassis; optionally one contact joint per tick on the chassis) and prints the worst deviation of position, rotation, linear
// and angular velocity over all bodies and ticks, relative to max(|ODE value|, 1e-3 * scale).  Exit code 0 iff all stay < 1e-3
// (two single-precision engines with different elimination orders: 1e-4 .. 1e-3 after a thousand ticks is agreement).
#include <ode/ode.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "pdb_types.h"
#include "../rb/pdrb.h"

static bool readFile(const char* path, void* dst, size_t n) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    const bool ok = fread(dst, 1, n, f) == n;
    fclose(f);
    return ok;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s <car.pdcar> <state.bin> [ticks=1000] [contacts=0|1]\n", argv[0]); return 2; }
    static pdb_car_params P; static pdb_dyn_state S;
    if (!readFile(argv[1], &P, sizeof(P)) || !readFile(argv[2], &S, sizeof(S))) { fprintf(stderr, "cannot read inputs\n"); return 2; }
    const int ticks = argc > 3 ? atoi(argv[3]) : 1000;
    const bool withContacts = argc > 4 && atoi(argv[4]) != 0;
    const float h = (float)(1.0 / 333.0);

    // ---- ODE side ----
    dInitODE2(0);
    dWorldID world = dWorldCreate();
    dWorldSetGravity(world, P.gravity[0], P.gravity[1], P.gravity[2]);
    dWorldSetERP(world, P.worldErp);
    dWorldSetCFM(world, P.worldCfm);
    dWorldSetContactMaxCorrectingVel(world, 3.0f);
    dWorldSetContactSurfaceLayer(world, 0.0f);
    dWorldSetDamping(world, 0.0f, 0.0f);
    dJointGroupID contactGroup = dJointGroupCreate(0);
    std::vector<dBodyID> ob(P.numBodies);
    // ---- pdrb side ----
    pdrb::World w;
    for (int k = 0; k < 3; ++k) w.gravity[k] = P.gravity[k];
    w.erp = P.worldErp; w.cfm = P.worldCfm;
    for (int i = 0; i < P.numBodies; ++i) {
        const pdb_body_state& bs = S.body[i];
        ob[i] = dBodyCreate(world);
        dBodySetFiniteRotationMode(ob[i], 1);                      // RigidBodyODE.cpp:15-16
        dMass m;
        dMassSetParameters(&m, P.bodies[i].mass, 0, 0, 0, P.bodies[i].inertia[0], P.bodies[i].inertia[1], P.bodies[i].inertia[2], 0, 0, 0);
        dBodySetMass(ob[i], &m);
        dBodySetPosition(ob[i], bs.pos[0], bs.pos[1], bs.pos[2]);
        dMatrix3 R = {bs.R[0], bs.R[1], bs.R[2], 0, bs.R[3], bs.R[4], bs.R[5], 0, bs.R[6], bs.R[7], bs.R[8], 0};
        dBodySetRotation(ob[i], R);
        dBodySetLinearVel(ob[i], bs.lvel[0], bs.lvel[1], bs.lvel[2]);
        dBodySetAngularVel(ob[i], bs.avel[0], bs.avel[1], bs.avel[2]);
        const int b = w.createBody();
        pdrb::Body& pb = w.bodies[b];
        pb.mass = P.bodies[i].mass; pb.invMass = 1.0f / pb.mass;
        for (int k = 0; k < 9; ++k) { pb.I[k] = 0; pb.invI[k] = 0; }
        for (int k = 0; k < 3; ++k) { pb.I[4 * k] = P.bodies[i].inertia[k]; pb.invI[4 * k] = 1.0f / P.bodies[i].inertia[k]; }
        pb.setPosition(bs.pos[0], bs.pos[1], bs.pos[2]);
        pb.setRotation(bs.R);
        memcpy(pb.lvel, bs.lvel, 12); memcpy(pb.avel, bs.avel, 12);
    }
    // joints: the block stores them in the solver's row order with body-local anchors; both engines get them through their
    // world-coordinate setters, in the same sequence
    for (int j = 0; j < P.numJoints; ++j) {
        const pdb_joint_def& jd = P.joints[j];
        const pdrb::Body &A = w.bodies[jd.b0], &B = w.bodies[jd.b1];
        float a1w[3], a2w[3], axw[3];
        A.relPointPos(jd.anchor1, a1w); B.relPointPos(jd.anchor2, a2w); A.vectorToWorld(jd.axis1, axw);
        dJointID oj = nullptr;
        switch (jd.type) {
        case PDB_JOINT_FIXED:
            oj = dJointCreateFixed(world, nullptr); dJointAttach(oj, ob[jd.b0], ob[jd.b1]); dJointSetFixed(oj);
            w.createFixed(jd.b0, jd.b1);
            break;
        case PDB_JOINT_BALL:
            oj = dJointCreateBall(world, nullptr); dJointAttach(oj, ob[jd.b0], ob[jd.b1]); dJointSetBallAnchor(oj, a1w[0], a1w[1], a1w[2]);
            w.createBall(jd.b0, jd.b1, a1w);
            break;
        case PDB_JOINT_SLIDER:
            oj = dJointCreateSlider(world, nullptr); dJointAttach(oj, ob[jd.b0], ob[jd.b1]); dJointSetSliderAxis(oj, axw[0], axw[1], axw[2]);
            w.createSlider(jd.b0, jd.b1, axw);
            break;
        default: {
            oj = dJointCreateDBall(world, nullptr); dJointAttach(oj, ob[jd.b0], ob[jd.b1]);
            dJointSetDBallAnchor1(oj, a1w[0], a1w[1], a1w[2]); dJointSetDBallAnchor2(oj, a2w[0], a2w[1], a2w[2]);
            dJointSetDBallDistance(oj, jd.distance);
            dJointSetDBallParam(oj, dParamERP, jd.erp); dJointSetDBallParam(oj, dParamCFM, jd.cfm);   // JointODE.cpp:68-75
            const int id = w.createDBall(jd.b0, jd.b1, a1w, a2w);
            w.joints[id].targetDistance = jd.distance; w.joints[id].erp = jd.erp; w.joints[id].cfm = jd.cfm;
            break; }
        }
    }

    double worst[4] = {0, 0, 0, 0};   // pos, R, lvel, avel
    const double scale[4] = {1.0, 1.0, 10.0, 1.0};
    for (int t = 0; t < ticks; ++t) {
        // identical scripted loads on both sides
        const float rock = 3000.0f * sinf(6.2831853f * (float)t / 333.0f);
        for (int i = 0; i < P.numBodies; ++i) {
            float f[3] = {0, 0, 0};
            const float* pos = w.bodies[i].pos;
            if (i == PDB_BODY_CHASSIS) { f[0] = rock; }
            else if (i != PDB_BODY_TANK) { const float pen = 0.30f - pos[1]; if (pen > 0.0f) f[1] = 150000.0f * pen - 3000.0f * w.bodies[i].lvel[1]; }
            const float at[3] = {pos[0], pos[1], pos[2]};
            w.bodies[i].addForceAtPos(f, at);
            const dReal* op = dBodyGetPosition(ob[i]);
            float fo[3] = {0, 0, 0};
            if (i == PDB_BODY_CHASSIS) fo[0] = rock;
            else if (i != PDB_BODY_TANK) { const float pen = 0.30f - (float)op[1]; if (pen > 0.0f) fo[1] = 150000.0f * pen - 3000.0f * (float)dBodyGetLinearVel(ob[i])[1]; }
            dBodyAddForceAtPos(ob[i], fo[0], fo[1], fo[2], op[0], op[1], op[2]);
        }
        w.contacts.clear();
        dJointGroupEmpty(contactGroup);
        if (withContacts && (t % 50) < 25) {   // a contact joint under the chassis, the two surface kinds of PhysicsEngineODE.cpp:295-322 in turn
            const int kind = (t / 50) & 1;
            pdrb::ContactJoint c; memset(&c, 0, sizeof(c));
            const float* cp = w.bodies[PDB_BODY_CHASSIS].pos;
            c.pos[0] = cp[0] + 0.4f; c.pos[1] = cp[1] - 0.3f; c.pos[2] = cp[2] + 1.0f; c.normal[1] = 1.0f; c.depth = 0.01f; c.kind = kind;
            w.contacts.push_back(c); w.contactBody = PDB_BODY_CHASSIS;
            dContact cj; memset(&cj, 0, sizeof(cj));
            const dReal* op = dBodyGetPosition(ob[PDB_BODY_CHASSIS]);
            cj.geom.pos[0] = op[0] + 0.4f; cj.geom.pos[1] = op[1] - 0.3f; cj.geom.pos[2] = op[2] + 1.0f; cj.geom.normal[1] = 1.0f; cj.geom.depth = 0.01f;
            if (kind == 0) { cj.surface.mode = 28692; cj.surface.mu = 0.25f; cj.surface.bounce = 0.01f; cj.surface.soft_cfm = 0.0001f; }
            else { cj.surface.mode = 28700; cj.surface.mu = 0.1f; cj.surface.bounce = 0; cj.surface.soft_cfm = 0.000952380942f; cj.surface.soft_erp = 0.714285731f; }
            dJointID j = dJointCreateContact(world, contactGroup, &cj);
            dJointAttach(j, ob[PDB_BODY_CHASSIS], nullptr);
        }
        w.step(h);
        dWorldStep(world, h);
        for (int i = 0; i < P.numBodies; ++i) {
            const pdrb::Body& pb = w.bodies[i];
            const dReal *op = dBodyGetPosition(ob[i]), *oR = dBodyGetRotation(ob[i]), *ov = dBodyGetLinearVel(ob[i]), *oa = dBodyGetAngularVel(ob[i]);
            double m[4] = {0, 0, 0, 0}, d[4] = {0, 0, 0, 0};
            for (int k = 0; k < 3; ++k) {
                d[0] = fmax(d[0], fabs(pb.pos[k] - op[k])); m[0] = fmax(m[0], fabs(op[k]));
                d[2] = fmax(d[2], fabs(pb.lvel[k] - ov[k])); m[2] = fmax(m[2], fabs(ov[k]));
                d[3] = fmax(d[3], fabs(pb.avel[k] - oa[k])); m[3] = fmax(m[3], fabs(oa[k]));
                for (int c = 0; c < 3; ++c) { d[1] = fmax(d[1], fabs(pb.R[3 * k + c] - oR[4 * k + c])); m[1] = 1.0; }
            }
            for (int q = 0; q < 4; ++q) worst[q] = fmax(worst[q], d[q] / fmax(m[q], 1e-3 * scale[q]));
        }
    }
    printf("pdrb vs ODE %s over %d ticks, %d bodies, %d joints%s: worst relative deviation pos %.3e  R %.3e  lvel %.3e  avel %.3e\n", dGetConfiguration(), ticks, P.numBodies,
           P.numJoints, withContacts ? ", contact joints" : "", worst[0], worst[1], worst[2], worst[3]);
    if (withContacts)   // where the restated contact solve is known to be able to differ from ODE's own
        printf("  expected source of deviation with contact joints: ODE's Dantzig routine fixes a friction row's limits from x[findex] when that row is first driven "
               "(earlier friction rows already in the solution); pdrb's two-stage scheme gives every friction row the frictionless normal force "
               "(the same LCP for the first friction row only)\n");
    const bool ok = worst[0] < 1e-3 && worst[1] < 1e-3 && worst[2] < 1e-3 && worst[3] < 1e-3;
    dJointGroupDestroy(contactGroup);
    dWorldDestroy(world);
    dCloseODE();
    return ok ? 0 : 1;
}
