// ORACLE / TEST INFRASTRUCTURE -- see cpu_ref.h.  Citations are relative to
// /root/reference/src/ProjectD unless stated otherwise.
#include "cpu_ref.h"
#include "../rb/pdcollide.h"
#include "mathsel.h"
#include "../rb/pdray.h"
#include <cmath>
#include <cstring>
#include <cfloat>
#include <algorithm>

namespace cpuref {

// ------------------------------------------------------------------------------------------------
// Core/Math.h:30-47,97-136 helpers with the reference's evaluation order
// ------------------------------------------------------------------------------------------------
struct V3 {
    float x = 0, y = 0, z = 0;
    V3() {}
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit V3(const float* p) : x(p[0]), y(p[1]), z(p[2]) {}
    float sqlen() const { return x * x + y * y + z * z; }
    float len() const { return sqrtf(sqlen()); }
    V3& norm(float l) { if (l != 0.0f) { const float s = 1.0f / l; x *= s; y *= s; z *= s; } return *this; }
    V3& norm() { return norm(len()); }
    V3 get_norm() const { V3 c = *this; c.norm(); return c; }
    V3 get_norm(float l) const { V3 c = *this; c.norm(l); return c; }
    V3 cross(const V3& v) const { return V3(y * v.z - z * v.y, z * v.x - x * v.z, x * v.y - y * v.x); }
    void store(float* p) const { p[0] = x; p[1] = y; p[2] = z; }
};
static inline V3 operator*(const V3& v, float f) { return V3(v.x * f, v.y * f, v.z * f); }
static inline V3 operator/(const V3& v, float f) { return V3(v.x / f, v.y / f, v.z / f); }
static inline V3 operator+(const V3& a, const V3& b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(const V3& a, const V3& b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline float operator*(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

template <typename T> static inline T tmin(T a, T b) { return a < b ? a : b; }
template <typename T> static inline T tmax(T a, T b) { return a > b ? a : b; }
template <typename T> static inline T tclamp(T x, T a, T b) { return x < a ? a : (x > b ? b : x); }
static inline float signf_(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
static inline float linscalef(float x, float x0, float x1, float r0, float r1) {
    x = tclamp(x, x0, x1);
    return ((r1 - r0) * (x - x0)) / (x1 - x0) + r0;
}

struct M44 {
    float m[16];  // M11..M44 row-major
    M44() { for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.0f : 0.0f; }
};
// Core/Math.cpp:87-115
static M44 axisAngle(const V3& a, float angle) {
    M44 r;
    const float s = m_sinf(angle), c = m_cosf(angle), o = 1.0f - c;
    r.m[0] = ((a.x * a.x) * o) + c; r.m[5] = ((a.y * a.y) * o) + c; r.m[10] = ((a.z * a.z) * o) + c;
    r.m[1] = (a.z * s) + (a.y * a.x) * o; r.m[6] = (a.x * s) + (a.z * a.y) * o; r.m[8] = (a.y * s) + (a.z * a.x) * o;
    r.m[2] = (a.z * a.x) * o - (a.y * s); r.m[4] = (a.y * a.x) * o - (a.z * s); r.m[9] = (a.z * a.y) * o - (a.x * s);
    r.m[3] = 0; r.m[7] = 0; r.m[11] = 0; r.m[12] = 0; r.m[13] = 0; r.m[14] = 0; r.m[15] = 1;
    return r;
}
// Core/Math.cpp:117-121 (XMMatrixMultiply; element order ((a0 b0 + a1 b1) + a2 b2) + a3 b3 -- project
// choice, DirectXMath's SIMD association is not reproducible here)
static M44 mult44(const M44& a, const M44& b) {
    M44 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r.m[i * 4 + j] = ((a.m[i * 4 + 0] * b.m[0 * 4 + j] + a.m[i * 4 + 1] * b.m[1 * 4 + j]) + a.m[i * 4 + 2] * b.m[2 * 4 + j]) + a.m[i * 4 + 3] * b.m[3 * 4 + j];
    return r;
}
// Core/Curve.cpp:94-115
static float curve(const pdb_curve& c, float ref) {
    if (c.n == 0) return 0.0f;
    if (ref <= c.x[0]) return c.y[0];
    for (int i = 1; i < c.n; ++i)
        if (ref <= c.x[i]) return (((c.y[i] - c.y[i - 1]) * (ref - c.x[i - 1])) / (c.x[i] - c.x[i - 1])) + c.y[i - 1];
    return c.y[c.n - 1];
}

// ------------------------------------------------------------------------------------------------
// IRigidBody accessors (Physics/ODE/RigidBodyODE.cpp:101-270)
// ------------------------------------------------------------------------------------------------
typedef pdrb::Body Body;
static V3 l2w(const Body& b, const V3& p) { float o[3]; b.relPointPos(&p.x, o); return V3(o); }
static V3 w2l(const Body& b, const V3& p) { float o[3]; b.posRelPoint(&p.x, o); return V3(o); }
static V3 l2wN(const Body& b, const V3& p) { float o[3]; b.vectorToWorld(&p.x, o); return V3(o); }
static V3 w2lN(const Body& b, const V3& p) { float o[3]; b.vectorFromWorld(&p.x, o); return V3(o); }
static V3 getVelocity(const Body& b) { const float z[3] = {0, 0, 0}; float o[3]; b.relPointVel(z, o); return V3(o); }
static V3 pointVel(const Body& b, const V3& p) { float o[3]; b.pointVel(&p.x, o); return V3(o); }
static V3 localPointVel(const Body& b, const V3& p) { float o[3]; b.relPointVel(&p.x, o); return V3(o); }
static V3 getPos(const Body& b) { return V3(b.pos); }
static M44 worldMatrix(const Body& b) {
    M44 m;
    const float* r = b.R;
    m.m[0] = r[0]; m.m[1] = r[3]; m.m[2] = r[6]; m.m[3] = 0;
    m.m[4] = r[1]; m.m[5] = r[4]; m.m[6] = r[7]; m.m[7] = 0;
    m.m[8] = r[2]; m.m[9] = r[5]; m.m[10] = r[8]; m.m[11] = 0;
    m.m[12] = b.pos[0]; m.m[13] = b.pos[1]; m.m[14] = b.pos[2]; m.m[15] = 1.0f;
    return m;
}

void TrackData::bind(const uint8_t* blob) {
    h = reinterpret_cast<const pdb_track_header*>(blob);
    surfaces = reinterpret_cast<const pdb_surface*>(blob + h->offSurfaces);
    tris = reinterpret_cast<const float*>(blob + h->offTris);
    fat = reinterpret_cast<const float*>(blob + h->offFat);
    fatDist = reinterpret_cast<const float*>(blob + h->offFatDist);
    nodes = reinterpret_cast<const float*>(blob + h->offNodes);
    nodeDist = reinterpret_cast<const float*>(blob + h->offNodeDist);
}

// ray vs track surfaces; same canonical algorithm as oracle/rb/pdray.h (kept on the blob layout)
struct Hit { bool has = false; float depth = -1; V3 pos, normal; int surface = -1; };
static Hit rayCast(const TrackData& T, const V3& o, const V3& d, float maxDist) {
    Hit best;
    for (int s = 0; s < T.h->numSurfaces; ++s) {
        float bt = -1.0f; int btri = -1;
        const int t0 = T.surfaces[s].triStart, t1 = t0 + T.surfaces[s].triCount;
        for (int t = t0; t < t1; ++t) {
            float tt;
            if (pdrb::rayTri(&o.x, &d.x, maxDist, T.tris + 9 * t, T.tris + 9 * t + 3, T.tris + 9 * t + 6, tt))
                if (bt < 0.0f || tt < bt) { bt = tt; btri = t; }
        }
        if (btri >= 0 && (best.depth < 0.0f || best.depth > bt)) {
            const float* v0 = T.tris + 9 * btri; const float* v1 = v0 + 3; const float* v2 = v0 + 6;
            const float vu[3] = {v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2]};
            const float vv[3] = {v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2]};
            float n[3] = {vu[1] * vv[2] - vu[2] * vv[1], vu[2] * vv[0] - vu[0] * vv[2], vu[0] * vv[1] - vu[1] * vv[0]};
            const float l = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
            if (l > 0.0f) {
                const float sc = 1.0f / sqrtf(l);
                best.has = true; best.depth = bt; best.surface = s;
                best.pos = V3(o.x + d.x * bt, o.y + d.y * bt, o.z + d.z * bt);
                best.normal = V3(n[0] * sc, n[1] * sc, n[2] * sc);
            }
        }
    }
    return best;
}

// ------------------------------------------------------------------------------------------------
void Car::init(const pdb_car_params* P_, const TrackData* T_, const pdb_dyn_state& s0) {
    P = P_; T = T_;
    memset(&slip, 0, sizeof(slip)); otherSlips.clear(); airDensityNow = P->airDensity;
    w = pdrb::World();
    w.erp = P->worldErp; w.cfm = P->worldCfm;
    for (int k = 0; k < 3; ++k) w.gravity[k] = P->gravity[k];
    for (int i = 0; i < P->numBodies; ++i) {
        const int id = w.createBody();
        Body& b = w.bodies[id];
        b.mass = P->bodies[i].mass; b.invMass = 1.0f / b.mass;
        for (int k = 0; k < 9; ++k) { b.I[k] = 0; b.invI[k] = 0; }
        for (int k = 0; k < 3; ++k) { b.I[k * 4] = P->bodies[i].inertia[k]; b.invI[k * 4] = 1.0f / P->bodies[i].inertia[k]; }
    }
    for (int j = 0; j < P->numJoints; ++j) {
        const pdb_joint_def& d = P->joints[j];
        pdrb::Joint jt;
        jt.type = d.type; jt.b0 = d.b0; jt.b1 = d.b1; jt.erp = d.erp; jt.cfm = d.cfm;
        memcpy(jt.anchor1, d.anchor1, 12); memcpy(jt.anchor2, d.anchor2, 12); memcpy(jt.axis1, d.axis1, 12);
        memcpy(jt.offset, d.offset, 12); memcpy(jt.qrel, d.qrel, 16);
        jt.targetDistance = d.distance;
        w.joints.push_back(jt);
    }
    w.jointOrder.resize(P->numJoints);
    for (int j = 0; j < P->numJoints; ++j) w.jointOrder[j] = j;   // params are already in solver order
    w.orderDirty = false;
    memset(&controls, 0, sizeof(controls));
    controls.isShifterSupported = 1; controls.requestedGearIndex = -1;
    for (int i = 0; i < 7; ++i) probeHits[i] = 0;
    for (int i = 0; i < 5; ++i) lookAhead[i] = 0;
    for (int i = 0; i < 4; ++i) { ts[i] = TyreScratch(); for (int k = 0; k < 16; ++k) ts[i].hubMatrix[k] = 0; }
    loadState(s0);
}

void Car::loadState(const pdb_dyn_state& s) {
    S = s;
    for (int i = 0; i < P->numBodies; ++i) {
        Body& b = w.bodies[i];
        memcpy(b.pos, s.body[i].pos, 12); memcpy(b.q, s.body[i].q, 16); memcpy(b.R, s.body[i].R, 36);
        memcpy(b.lvel, s.body[i].lvel, 12); memcpy(b.avel, s.body[i].avel, 12);
        for (int k = 0; k < 3; ++k) { b.facc[k] = 0; b.tacc[k] = 0; }
    }
    // Track::nearbyPoints is a pure function of pointCachePos (refresh radius = probe 0's length)
    nearby.clear();
    const float md = P->probeLen[0], md2 = md * md;
    const V3 cp(S.pointCachePos);
    for (int id = 0; id < T->h->numFat; ++id) { const V3 p(T->fat + 15 * id); if ((cp - p).sqlen() < md2) nearby.push_back(id); }
    stepTime = S.physicsTime;
    locClutch = S.locClutch;
}
void Car::setContacts(const pdb_contact* c, int n) {
    contactSet.clear();
    if (n > pdcol::MAX_CONTACTS) n = pdcol::MAX_CONTACTS;
    for (int i = 0; i < n; ++i) { memcpy(&contactSet.c[i], &c[i], sizeof(pdb_contact)); contactSet.id[i] = 0; }
    contactSet.n = n;
    S.numContacts = n;
}
void Car::getContacts(pdb_contact* c) const {
    memset(c, 0, sizeof(pdb_contact) * PDB_MAX_CONTACTS);
    for (int i = 0; i < contactSet.n && i < S.numContacts; ++i) memcpy(&c[i], &contactSet.c[i], sizeof(pdb_contact));
}
void Car::storeState() {
    for (int i = 0; i < P->numBodies; ++i) {
        const Body& b = w.bodies[i];
        memcpy(S.body[i].pos, b.pos, 12); memcpy(S.body[i].q, b.q, 16); memcpy(S.body[i].R, b.R, 36);
        memcpy(S.body[i].lvel, b.lvel, 12); memcpy(S.body[i].avel, b.avel, 12);
    }
}

// ------------------------------------------------------------------------------------------------
// suspension helpers
// ------------------------------------------------------------------------------------------------
// Damper::getForce (Car/Damper.cpp:11-29)
static float damperForce(const pdb_damper& d, float speed) {
    float f;
    if (speed <= 0.0f) {
        if (fabsf(speed) <= d.fastThresholdRebound) f = -(speed * d.reboundSlow);
        else f = (d.fastThresholdRebound * d.reboundSlow) - ((d.fastThresholdRebound + speed) * d.reboundFast);
    } else {
        if (speed <= d.fastThresholdBump) f = -(speed * d.bumpSlow);
        else f = -(((speed - d.fastThresholdBump) * d.bumpFast) + (d.fastThresholdBump * d.bumpSlow));
    }
    return f;
}

// ISuspension::getHubWorldMatrix (SuspensionStrut.cpp:367-373, SuspensionAxle.cpp:224-232)
static M44 hubWorldMatrix(const pdb_car_params& P, const pdrb::World& w, int i) {
    const pdb_susp& su = P.susp[i];
    if (su.type != PDB_SUSP_AXLE) {   // SuspensionDW.cpp:343-346, SuspensionML.cpp:198-201: mat44f::rotate == the same product
        const M44 m0 = worldMatrix(w.bodies[su.hubBody]);
        const M44 rot = axisAngle(V3(0, 0, 1), su.staticCamber);
        return mult44(rot, m0);
    }
    M44 m = worldMatrix(w.bodies[su.hubBody]);
    const float t = (su.sideSign < 0.0f) ? -su.axleTrack : su.axleTrack;
    m.m[12] += m.m[0] * t; m.m[13] += m.m[1] * t; m.m[14] += m.m[2] * t;
    return m;
}

// SuspensionStrut::step (SuspensionStrut.cpp:230-290)
static void strutStep(const pdb_susp& su, pdrb::World& w, TyreScratch& sc) {
    Body& body = w.bodies[PDB_BODY_CHASSIS];
    Body& hub = w.bodies[su.hubBody];
    const M44 mb = worldMatrix(body);
    const V3 vCarStrut = l2w(body, V3(su.carStrut));
    const V3 vTyreStrut = l2w(hub, V3(su.tyreStrut));
    V3 vDelta = vTyreStrut - vCarStrut;
    const float fDeltaLen = vDelta.len();
    vDelta.norm(fDeltaLen);
    const float fDefaultLength = su.strutBaseLength + su.rodLength;
    const float fTravel = fDefaultLength - fDeltaLen;
    sc.travel = fTravel;
    float fForce = ((fTravel * su.progressiveK) + su.k) * fTravel;
    if (fForce < 0) fForce = 0;
    if (su.packerRange != 0.0f && fTravel > su.packerRange) fForce += ((fTravel - su.packerRange) * su.bumpStopRate);
    if (fForce > 0) {
        const V3 vForce = vDelta * fForce;
        hub.addForceAtPos(&vForce.x, &vTyreStrut.x);
        const V3 neg = vForce * -1.0f;
        body.addForceAtPos(&neg.x, &vCarStrut.x);
    }
    const V3 vHubWorld = getPos(hub);
    const V3 vHubLocal = w2l(body, vHubWorld);
    const float fHubDelta = vHubLocal.y - su.refPointY;
    const V3 m2(mb.m[4], mb.m[5], mb.m[6]);
    if (fHubDelta > su.bumpStopUp) {
        fForce = (fHubDelta - su.bumpStopUp) * 500000.0f;
        const V3 f = m2 * -fForce; const V3 p = getPos(hub);
        hub.addForceAtPos(&f.x, &p.x);
        const V3 lf(0, fForce, 0);
        body.addRelForceAtRelPos(&lf.x, &vHubLocal.x);
    }
    if (fHubDelta < su.bumpStopDn) {
        fForce = (fHubDelta - su.bumpStopDn) * 500000.0f;
        const V3 f = m2 * -fForce; const V3 p = getPos(hub);
        hub.addForceAtPos(&f.x, &p.x);
        const V3 lf(0, fForce, 0);
        body.addRelForceAtRelPos(&lf.x, &vHubLocal.x);
    }
    const V3 vTyreStrutVel = localPointVel(hub, V3(su.tyreStrut));
    const V3 vCarStrutVel = localPointVel(body, V3(su.carStrut));
    const V3 vDamperDelta = vTyreStrutVel - vCarStrutVel;
    const float fDamperSpeed = vDamperDelta * vDelta;
    sc.damperSpeedMS = fDamperSpeed;
    const float fDamperForce = damperForce(su.damper, fDamperSpeed);
    const V3 vDamperForce = vDelta * fDamperForce;
    hub.addForceAtPos(&vDamperForce.x, &vTyreStrut.x);
    const V3 negd = vDamperForce * -1.0f;
    body.addForceAtPos(&negd.x, &vCarStrut.x);
}

// SuspensionDW::step (SuspensionDW.cpp:214-283), no active actuator
static void dwStep(const pdb_susp& su, pdrb::World& w, TyreScratch& sc) {
    Body& body = w.bodies[PDB_BODY_CHASSIS];
    Body& hub = w.bodies[su.hubBody];
    const M44 mb = worldMatrix(body);
    const V3 vBodyM2(mb.m[4], mb.m[5], mb.m[6]);
    const V3 vHubWorldPos = getPos(hub);
    const V3 vHubLocalPos = w2l(body, vHubWorldPos);
    const V3 refPoint(su.basePosition);
    const float fHubDeltaY = vHubLocalPos.y - refPoint.y;
    const float fTravel = fHubDeltaY + su.rodLength;
    sc.travel = fTravel;
    float fForce = ((fTravel * su.progressiveK) + su.k) * fTravel;
    if (su.packerRange != 0.0f && fTravel > su.packerRange && su.k != 0.0f)
        fForce += (((fTravel - su.packerRange) * su.bumpStopProgressive) + su.bumpStopRate) * (fTravel - su.packerRange);
    if (fForce > 0.0f) {
        const V3 f = vBodyM2 * -fForce;
        hub.addForceAtPos(&f.x, &vHubWorldPos.x);
        const V3 lf(0, fForce, 0);
        body.addRelForceAtRelPos(&lf.x, &refPoint.x);
    }
    const V3 vHubVel = getVelocity(hub);
    const V3 vPointVel = localPointVel(body, refPoint);
    const V3 vDeltaVel = vHubVel - vPointVel;
    const float fDamperSpeed = vDeltaVel * vBodyM2;
    sc.damperSpeedMS = fDamperSpeed;
    const float fDamperForce = damperForce(su.damper, fDamperSpeed);
    {
        const V3 vForce = vBodyM2 * fDamperForce;
        hub.addForceAtPos(&vForce.x, &vHubWorldPos.x);
        const V3 neg = vForce * -1.0f;
        body.addForceAtRelPos(&neg.x, &refPoint.x);
    }
    if (su.bumpStopUp != 0.0f && fHubDeltaY > su.bumpStopUp && 0.0f != su.k) {
        fForce = (((fHubDeltaY - su.bumpStopUp) * su.bumpStopProgressive) + su.bumpStopRate) * (fHubDeltaY - su.bumpStopUp);
        const V3 f = vBodyM2 * -fForce;
        hub.addForceAtPos(&f.x, &vHubWorldPos.x);
        const V3 lf(0, fForce, 0);
        body.addRelForceAtRelPos(&lf.x, &vHubLocalPos.x);
    }
    if (su.bumpStopDn != 0.0f && fHubDeltaY < su.bumpStopDn && 0.0f != su.k) {
        fForce = (((fHubDeltaY - su.bumpStopDn) * su.bumpStopProgressive) + su.bumpStopRate) * (fHubDeltaY - su.bumpStopDn);
        const V3 f = vBodyM2 * -fForce;
        hub.addForceAtPos(&f.x, &vHubWorldPos.x);
        const V3 lf(0, fForce, 0);
        body.addRelForceAtRelPos(&lf.x, &vHubLocalPos.x);
    }
}

// SuspensionML::step (SuspensionML.cpp:106-137): like the double wishbone's travel spring, but applied whatever its sign,
// a plain packer term and no bump stops
static void mlStep(const pdb_susp& su, pdrb::World& w, TyreScratch& sc) {
    Body& body = w.bodies[PDB_BODY_CHASSIS];
    Body& hub = w.bodies[su.hubBody];
    const M44 mb = worldMatrix(body);
    const V3 vM2(mb.m[4], mb.m[5], mb.m[6]);
    const V3 vHubWorld = getPos(hub);
    const V3 vHubLocal = w2l(body, vHubWorld);
    const V3 basePos(su.basePosition);
    const float fTravel = (vHubLocal.y - basePos.y) + su.rodLength;
    sc.travel = fTravel;
    float fForce = (fTravel * su.progressiveK + su.k) * fTravel;
    if (su.packerRange != 0.0f && fTravel > su.packerRange && su.k != 0.0f) fForce += ((fTravel - su.packerRange) * su.bumpStopRate);
    {
        const V3 f = vM2 * -fForce;
        hub.addForceAtPos(&f.x, &vHubWorld.x);
        const V3 lf(0.0f, fForce, 0.0f);
        body.addRelForceAtRelPos(&lf.x, &basePos.x);
    }
    const V3 vPointVel = localPointVel(body, basePos);
    const V3 vDeltaVel = getVelocity(hub) - vPointVel;
    const float fDamperSpeed = vDeltaVel * vM2;
    sc.damperSpeedMS = fDamperSpeed;
    const V3 vForce = vM2 * damperForce(su.damper, fDamperSpeed);
    hub.addForceAtPos(&vForce.x, &vHubWorld.x);
    const V3 neg = vForce * -1.0f;
    body.addForceAtRelPos(&neg.x, &basePos.x);
}

// HeaveSpring::step (HeaveSpring.cpp:56-149) for the axle whose wheels are i0, i0 + 1 (both double wishbones)
static void heaveStep(const pdb_car_params& P, const pdb_heave& H, pdrb::World& w, int i0) {
    const pdb_susp& s0 = P.susp[i0];
    const pdb_susp& s1 = P.susp[i0 + 1];
    Body& body = w.bodies[PDB_BODY_CHASSIS];
    Body& hub0 = w.bodies[s0.hubBody];
    Body& hub1 = w.bodies[s1.hubBody];
    const M44 mb = worldMatrix(body);
    const V3 vM2(mb.m[4], mb.m[5], mb.m[6]);
    const V3 ref0(s0.basePosition), ref1(s1.basePosition);
    const V3 hubPos0 = getPos(hub0), hubPos1 = getPos(hub1);
    const V3 hubLoc0 = w2l(body, hubPos0), hubLoc1 = w2l(body, hubPos1);
    float rodLength = H.rodLength;
    if (s0.k != 0.0f || s1.k != 0.0f) rodLength = (s1.rodLength + s0.rodLength) * 0.5f;
    const float fAvgY = (hubLoc0.y + hubLoc1.y) * 0.5f;
    const float fTravel = (fAvgY - ref0.y) + rodLength;
    float v12 = ((fTravel * H.progressiveK) + H.k) * fTravel;
    if (H.packerRange != 0.0f && fTravel > H.packerRange) v12 += ((fTravel - H.packerRange) * H.bumpStopRate);
    auto pair = [&](const V3& hubForce, const V3& bodyLocalForce) {
        hub0.addForceAtPos(&hubForce.x, &hubPos0.x);
        hub1.addForceAtPos(&hubForce.x, &hubPos1.x);
        body.addRelForceAtRelPos(&bodyLocalForce.x, &ref0.x);
        body.addRelForceAtRelPos(&bodyLocalForce.x, &ref1.x);
    };
    pair(vM2 * -v12, V3(0, v12, 0));
    const float fDeltaY0 = fAvgY - ref0.y;
    if (H.bumpStopUp != 0.0f && fDeltaY0 > H.bumpStopUp) { const float f = (fDeltaY0 - H.bumpStopUp) * 500000.0f; pair(vM2 * -f, V3(0, f, 0)); }
    if (H.bumpStopDn != 0.0f && fDeltaY0 < H.bumpStopDn) { const float f = (fDeltaY0 - H.bumpStopDn) * 500000.0f; pair(vM2 * -f, V3(0, f, 0)); }
    const V3 vHubVel = (getVelocity(hub0) + getVelocity(hub1)) * 0.5f;
    const V3 vLpv = (localPointVel(body, ref0) + localPointVel(body, ref1)) * 0.5f;
    const V3 vDeltaVel = vHubVel - vLpv;
    const float fDamperSpeed = vDeltaVel * vM2;
    const V3 vForce = vM2 * damperForce(H.damper, fDamperSpeed);
    pair(vForce, vForce * -1.0f);   // the reference hands the world-space vector to addLocalForceAtLocalPos (HeaveSpring.cpp:145-147)
}

// SuspensionAxle::step (SuspensionAxle.cpp:120-185)
static void axleStep(const pdb_susp& su, pdrb::World& w, TyreScratch& sc) {
    Body& body = w.bodies[PDB_BODY_CHASSIS];
    Body& axle = w.bodies[su.hubBody];
    const M44 mb = worldMatrix(body);
    const M44 ma = worldMatrix(axle);
    const V3 vAxleM1(ma.m[0], ma.m[1], ma.m[2]);
    const V3 vAxleM4(ma.m[12], ma.m[13], ma.m[14]);
    const float fSideSign = (su.sideSign < 0.0f) ? -1.0f : 1.0f;
    const V3 vAxleWorld = (vAxleM1 * (fSideSign * su.axleTrack * su.attachRelativePos)) + vAxleM4;
    const V3 vAxleLocal = w2l(body, vAxleWorld);
    V3 vBase(su.basePosition);
    vBase.x *= su.attachRelativePos;
    vBase.y += 0.2f;
    const V3 vBaseWorld = l2w(body, vBase);
    V3 vDelta = vBaseWorld - vAxleWorld;
    const float fDeltaLen = vDelta.len();
    vDelta.norm(fDeltaLen);
    const float fTravel = (0.2f - fDeltaLen) + su.rodLength;
    sc.travel = fTravel;
    float fForce = -(((fTravel * su.progressiveK) + su.k) * fTravel);
    if (fForce < 0.0f) {
        const V3 f = vDelta * fForce;
        axle.addForceAtPos(&f.x, &vAxleWorld.x);
        const V3 g = vDelta * -fForce;
        body.addForceAtPos(&g.x, &vBaseWorld.x);
    }
    const V3 m1(mb.m[0], mb.m[1], mb.m[2]);
    const V3 m2(mb.m[4], mb.m[5], mb.m[6]);
    if (su.leafSpringKx != 0.0f) {
        fForce = (vAxleLocal.x - (su.attachRelativePos * su.basePosition[0])) * su.leafSpringKx;
        const V3 f = m1 * -fForce;
        axle.addForceAtPos(&f.x, &vAxleWorld.x);
        const V3 lf(fForce, 0.0f, 0.0f);
        body.addRelForceAtRelPos(&lf.x, &vAxleLocal.x);
    }
    const float fRefY = vAxleLocal.y - su.referenceY;
    if (su.bumpStopUp != 0.0f && fRefY > su.bumpStopUp && 0.0f != su.k) {
        fForce = (fRefY - su.bumpStopUp) * 500000.0f;
        const V3 f = m2 * -fForce;
        axle.addForceAtPos(&f.x, &vAxleWorld.x);
        const V3 lf(0.0f, fForce, 0.0f);
        body.addRelForceAtRelPos(&lf.x, &vAxleLocal.x);
    }
    if (su.bumpStopDn != 0.0f && fRefY < su.bumpStopDn && 0.0f != su.k) {
        fForce = (fRefY - su.bumpStopDn) * 500000.0f;
        const V3 f = m2 * -fForce;
        axle.addForceAtPos(&f.x, &vAxleWorld.x);
        const V3 lf(0.0f, fForce, 0.0f);
        body.addRelForceAtRelPos(&lf.x, &vAxleLocal.x);
    }
    const V3 vPointVel = pointVel(body, vBaseWorld);
    const float t = (su.sideSign < 0.0f) ? -su.axleTrack : su.axleTrack;
    const V3 vDeltaVel = localPointVel(axle, V3(t, 0, 0)) - vPointVel;
    const float fDamperSpeed = vDeltaVel * vDelta;
    sc.damperSpeedMS = fDamperSpeed;
    const float fDamperForce = damperForce(su.damper, fDamperSpeed);
    const V3 vForce = vDelta * fDamperForce;
    axle.addForceAtPos(&vForce.x, &vAxleWorld.x);
    const V3 neg = vForce * -1.0f;
    body.addForceAtPos(&neg.x, &vBaseWorld.x);
}

// ------------------------------------------------------------------------------------------------
// SCTM tyre model (Car/TyreModel.cpp:11-161)
// ------------------------------------------------------------------------------------------------
struct TMI { float load, slipAngleRAD, slipRatio, camberRAD, speed, u, cpLength, grain, blister, pressureRatio; bool useSimpleModel; };
struct TMO { float Fy = 0, Fx = 0, Mz = 0, trail = 0, ndSlip = 0, Dy = 0, Dx = 0; };

// Curve::getCubicSplineValue (Core/Curve.cpp:117-126): the vendored tk::spline's operator() in float -- the last point below x (the first one for
// anything left of the points), the cubic of that interval, quadratic continuation outside
static float splineValue(const pdb_spline& s, float x) {
    int lb = 0;
    while (lb < s.n && s.x[lb] < x) ++lb;          // std::lower_bound
    const int idx = lb - 1 > 0 ? lb - 1 : 0;
    const float h = x - s.x[idx];
    if (x < s.x[0]) return (s.b0 * h + s.c0) * h + s.y[0];
    if (x > s.x[s.n - 1]) return (s.b[s.n - 1] * h + s.c[s.n - 1]) * h + s.y[s.n - 1];
    return ((s.a[idx] * h + s.b[idx]) * h + s.c[idx]) * h + s.y[idx];
}
// Curve::getValue (Core/Curve.cpp:94-115) on a spline's points
static float splineLinear(const pdb_spline& s, float ref) {
    if (s.n <= 0) return 0.0f;
    if (ref <= s.x[0]) return s.y[0];
    for (int id = 1; id < s.n; ++id) if (ref <= s.x[id]) return (((s.y[id] - s.y[id - 1]) * (ref - s.x[id - 1])) / (s.x[id] - s.x[id - 1])) + s.y[id - 1];
    return s.y[s.n - 1];
}
// SCTM::getStaticDX / getStaticDY (TyreModel.cpp:121-146)
static float sctmStaticDX(const pdb_tyre& t, float load) { if (t.curveFlags & 2) return splineValue(t.dxLoadCurve, load); if (load != 0.0) return (m_powf(load, t.lsExpX) * t.lsMultX) / load; return 0; }
static float sctmStaticDY(const pdb_tyre& t, float load) { if (t.curveFlags & 1) return splineValue(t.dyLoadCurve, load); if (load != 0.0f) return (m_powf(load, t.lsExpY) * t.lsMultY) / load; return 0; }
static float sctmPureFY(const pdb_tyre& t, float asy, float /*D*/, float cf, float /*load*/, float slip) {
    const float v5 = (cf * 2.0f) * 0.0064f;
    const float v6 = 1.0f / (v5 / 3.0f);
    float fy;
    if (v6 < slip) fy = ((1.0f / (((slip - v6) * t.falloffSpeed) + 1.0f)) * (1.0f - asy)) + asy;
    else fy = (((1.0f - (slip / v6)) * (1.0f - (slip / v6))) * (v5 * slip)) + ((3.0f - ((slip / v6) * 2.0f)) * ((slip / v6) * (slip / v6)));
    return fy;
}
static TMO sctmSolve(const pdb_tyre& t, const TMI& tmi) {
    TMO tmo;
    if (tmi.load <= 0.0f || (tmi.slipAngleRAD == 0.0f && tmi.slipRatio == 0.0f && tmi.camberRAD == 0.0f)) return tmo;
    const float asy = tmi.useSimpleModel ? 1.0f : t.asy;
    const float fSlipAngle = tmi.slipAngleRAD;
    const float fUnk1 = (m_sinf(tmi.camberRAD) * t.camberGain) + fSlipAngle;
    const float fUnk1Tan = m_tanf(fUnk1);
    const float fSlipAngleSin = m_sinf(fSlipAngle);
    const float fBlister1 = tclamp(tmi.blister * 0.01f, 0.0f, 1.0f);
    const float fBlister2 = (fBlister1 * 0.2f) + 1.0f;
    const float fStaticDy = sctmStaticDY(t, tmi.load);
    const float fStaticDx = sctmStaticDX(t, tmi.load);
    float fUDy = tmi.u * fStaticDy / fBlister2;
    float fUDx = tmi.u * fStaticDx / fBlister2;
    if (tmi.slipRatio < 0.0f) fUDx = fUDx * t.brakeDXMod;
    const float fCamberRad = tmi.camberRAD;
    float fCamberRadTmp = fabsf(fCamberRad);
    if ((fCamberRad < 0.0f || fUnk1 < 0.0f) && (fCamberRad > 0.0f || fUnk1 > 0.0f)) fCamberRadTmp = -fCamberRadTmp;
    fCamberRadTmp = -fCamberRadTmp;
    if (t.curveFlags & 4) {   // DCAMBER_LUT (TyreModel.cpp:49-57)
        const float fCamberDeg = fCamberRadTmp * 57.29578f;
        if (t.curveFlags & 8) fUDy *= splineValue(t.dCamberCurve, fCamberDeg);
        else fUDy *= splineLinear(t.dCamberCurve, fCamberDeg);
    } else {
        float fCamberUnk = (fCamberRadTmp * t.dcamber0) - ((fCamberRadTmp * fCamberRadTmp) * t.dcamber1);
        if (fCamberUnk <= -1.0f) fCamberUnk = -0.8999999f;
        fUDy += (((fUDy / (fCamberUnk + 1.0f)) - fUDy) * t.dCamberBlend);
    }
    const float fSlipRatio = tmi.slipRatio;
    const float fSlipAngleCos = m_cosf(tmi.slipAngleRAD);
    const float fSlipRatioClamped = (fSlipRatio > -0.9999999f ? fSlipRatio : -0.9999999f);
    const float fSpeed = tmi.speed;
    const float a = fSpeed * fSlipAngleSin;
    const float b = (fSpeed * fSlipRatio) * fSlipAngleCos;
    const float fUnk2 = sqrtf((a * a) + (b * b));
    const float fUnk2Scaled = fUnk2 * t.speedSensitivity;
    const float fDy = fUDy / (fUnk2Scaled + 1.0f);
    const float fDx = fUDx / (fUnk2Scaled + 1.0f);
    const float fLoadSubFz0 = tmi.load - t.Fz0;
    const float fPCfGain = t.pressureCfGain;
    const float fCF = ((((1.0f / ((((fLoadSubFz0 / t.Fz0) * (t.maxSlip1 - t.maxSlip0)) + t.maxSlip0) * (((tmi.u - 1.0f) * 0.75f) + 1.0f))) * 3.0f) * 78.125f) / ((tmi.grain * 0.01f) + 1.0f)) * ((fPCfGain * tmi.pressureRatio) + 1.0f);
    const float fUnk3 = fSlipRatio / (fSlipRatioClamped + 1.0f);
    const float fUnk4 = fUnk1Tan / (fSlipRatioClamped + 1.0f);
    float fSlip;
    const float fCombFactor = t.combinedFactor;
    if (fCombFactor <= 0.0f || fCombFactor == 2.0f) fSlip = sqrtf((fUnk4 * fUnk4) + (fUnk3 * fUnk3));
    else {
        const float c = m_powf(fabsf(fUnk4), fCombFactor) + m_powf(fabsf(fUnk3), fCombFactor);
        fSlip = m_powf(c, 1.0f / fCombFactor);
    }
    const float fPureFyDx = sctmPureFY(t, asy, fDx, fCF * t.cfXmult, tmi.load, fSlip) * fDx;
    const float fPureFyDy = sctmPureFY(t, asy, fDy, fCF, tmi.load, fSlip);
    tmo.Fy = ((fPureFyDy * fDy) * (fUnk4 / fSlip)) * tmi.load;
    tmo.Fx = ((fUnk3 / fSlip) * fPureFyDx) * tmi.load;
    const float fNdSlip = fSlip / (1.0f / (((fCF * 2.0f) * 0.0064f) / 3.0f));
    const float fUnk5 = tclamp((1.0f - (fNdSlip * 0.8f)), 0.0f, 1.0f);
    const float fUnk6 = (((((3.0f - (fUnk5 * 2.0f)) * (fUnk5 * fUnk5)) * 1.1f) - 0.1f) * tmi.cpLength) * 0.12f;
    tmo.Mz = -(fUnk6 * tmo.Fy);
    tmo.trail = fUnk6 * tclamp(tmi.speed, 0.0f, 1.0f);
    tmo.ndSlip = fNdSlip;
    tmo.Dy = fDy;
    tmo.Dx = fDx;
    return tmo;
}

// ------------------------------------------------------------------------------------------------
// tyre thermal model (Car/TyreThermalModel.cpp:60-166)
// ------------------------------------------------------------------------------------------------
static int thermalElem(double phase) {
    const float fPhase = (float)(phase * 0.1591549430964443);
    return ((int)(fPhase * 12)) % 12;
}
static void thermalAddInput(const pdb_car_params& P, const pdb_tyre& tp, const pdb_tyre_state& st, float* inputT, float xpos, float pressureRel, float temp) {
    const float fNormXcs = tclamp((xpos * tp.camberSpreadK), -1.0f, 1.0f);
    const int e = thermalElem(st.phase);
    const float fT = P.roadTemperature + temp;
    const float fPr1 = pressureRel * 0.1f;
    const float fPr2 = (pressureRel * -0.5f) + 1.0f;
    inputT[e + 0 * 12] += ((((fNormXcs + 1.0f) - (fPr1 * 0.5f)) * fPr2) * fT);
    inputT[e + 1 * 12] += (((fPr1 + 1.0f) * fPr2) * fT);
    inputT[e + 2 * 12] += ((((1.0f - fNormXcs) - (fPr1 * 0.5f)) * fPr2) * fT);
}
static float thermalCPTemp(const pdb_tyre& tp, const pdb_tyre_state& st, float camber) {
    const float fNormCsk = tclamp((camber * tp.camberSpreadK), -1.0f, 1.0f);
    const int e = thermalElem(st.phase);
    return ((((fNormCsk + 1.0f) * st.T[e]) + st.T[e + 12]) + ((1.0f - fNormCsk) * st.T[e + 24])) * 0.33333334f;
}
static void thermalStep(const pdb_car_params& P, const pdb_tyre& tp, pdb_tyre_state& st, float* inputT, float& coreTInput, float dt, float angularSpeed, float camberRAD, float carSpeed) {
    float fPhase = (float)st.phase + (angularSpeed * dt);
    if (fPhase > 100000.0) fPhase -= 100000.0;
    else if (fPhase < 0.0) fPhase += 100000.0;
    st.phase = fPhase;
    const float fAmbientTemp = P.ambientTemperature;
    const float fCoreTempInput = tmax(fAmbientTemp, coreTInput);
    st.coreTemp += ((fCoreTempInput - st.coreTemp) * (tp.internalCoreTransfer * dt));
    coreTInput = 0;
    const float fSpeed = carSpeed;
    const float fAmbientFactor = ((((fSpeed * fSpeed) * tp.coolFactorGain) + 1.0f) * tp.surfaceTransfer) * dt;
    const float fPctDt = tp.patchCoreTransfer * dt;
    for (int k = 0; k < 36; ++k) {
        const float fInputT = inputT[k];
        float fPatchT = st.T[k];
        if (fInputT <= fAmbientTemp) fPatchT += ((fAmbientTemp - fPatchT) * fAmbientFactor);
        else fPatchT += ((fInputT - fPatchT) * (tp.surfaceTransfer * dt));
        for (int c = 0; c < P.patchConnCount[k]; ++c) fPatchT += (st.T[(int)P.patchConn[k][c]] - fPatchT) * (tp.patchTransfer * dt);
        fPatchT += (st.coreTemp - fPatchT) * fPctDt;
        st.T[k] = fPatchT;
        inputT[k] = 0;
        st.coreTemp += ((fPatchT - st.coreTemp) * fPctDt);
    }
    if (tp.performanceCurve.n > 0) {
        const float fPracT = ((thermalCPTemp(tp, st, camberRAD) - st.coreTemp) * 0.25f) + st.coreTemp;
        st.practicalTemp = fPracT;
        st.thermalMultD = curve(tp.performanceCurve, fPracT);
    }
}

// TyreUtils.inl:7-30
static float calcSlipAngleRAD(float vy, float vx) { if (vx != 0.0f) return m_atanf(-(vy / fabsf(vx))); return 0; }
static float calcCamberRAD(const V3& n, const M44& m) {
    const float f = ((m.m[1] * n.y) + (m.m[0] * n.x)) + (m.m[2] * n.z);
    if (f <= -1.0f || f >= 1.0f) return -1.5707964f;
    return -m_asinf(f);
}
static float calcContactPatchLength(float radius, float deflection) {
    const float v = radius - deflection;
    if (v <= 0.0f || radius <= v) return 0.0f;
    return sqrtf((radius * radius) - (v * v)) * 2.0f;
}

// ------------------------------------------------------------------------------------------------
// Tyre::step (Car/Tyre.cpp:427-659) with addGroundContact (:661-723), addTyreForcesV10
// (Car/TyreForces.cpp:13-194), getCorrectedD/stepDirtyLevel/stepPuncture (:196-247),
// updateLockedState/updateAngularSpeed (:725-752), stepThermalModel (:766-815), stepFlatSpot (:924-950)
// ------------------------------------------------------------------------------------------------
static void hubAddForceAtPos(pdrb::World& w, const pdb_susp& su, const V3& f, const V3& p) { w.bodies[su.hubBody].addForceAtPos(&f.x, &p.x); }
static void hubAddTorque(pdrb::World& w, const pdb_susp& su, const V3& t) { w.bodies[su.hubBody].addTorque(&t.x); }
static V3 hubPointVelocity(const pdrb::World& w, const pdb_susp& su, const V3& p) { return pointVel(w.bodies[su.hubBody], p); }

static void updateLockedState(const pdb_tyre& tp, pdb_tyre_state& st, const TyreScratch& sc) {
    if (st.isLocked) {
        const float fBrake = tmax(1.0f * sc.brakeTorque, sc.handBrakeTorque);
        st.isLocked = (fabsf(fBrake) >= fabsf(st.loadedRadius * st.Fx)) && (fabsf(st.angularVelocity) < 1.0f) && (!tp.driven);
    }
}

static void tyreStep(Car& c, int i, float dt) {
    const pdb_car_params& P = *c.P;
    const pdb_tyre& tp = P.tyre[i];
    const pdb_susp& su = P.susp[i];
    pdb_tyre_state& st = c.S.tyre[i];
    TyreScratch& sc = c.ts[i];
    pdrb::World& w = c.w;

    sc.feedbackTorque = 0; st.Fx = 0; st.Mz = 0; sc.slipFactor = 0; sc.rollingResistence = 0;
    sc.slidingVelocityY = 0; sc.slidingVelocityX = 0; sc.totalHubVelocity = 0; sc.surface = -1;
    const M44 mxWorld = hubWorldMatrix(P, w, i);
    const V3 vWorldM2(mxWorld.m[4], mxWorld.m[5], mxWorld.m[6]);
    const V3 worldPosition(mxWorld.m[12], mxWorld.m[13], mxWorld.m[14]);
    M44 worldRotation = mxWorld;
    worldRotation.m[12] = 0; worldRotation.m[13] = 0; worldRotation.m[14] = 0;
    if (!std::isfinite(st.angularVelocity)) st.angularVelocity = 0;

    V3 vHitPos(0, 0, 0), vHitNorm(0, 0, 0);
    const V3 vRayPos(worldPosition.x, worldPosition.y + 2.0f, worldPosition.z);
    const Hit hit = rayCast(*c.T, vRayPos, V3(0.0f, -1.0f, 0.0f), 3.0f);
    const bool bHasContact = hit.has;
    const pdb_surface* surf = nullptr;
    if (bHasContact) { vHitPos = hit.pos; vHitNorm = hit.normal; surf = &c.T->surfaces[hit.surface]; }
    float inputT[36];
    for (int k = 0; k < 36; ++k) inputT[k] = st.inputT0;
    float coreTInput = 0;

    if (!bHasContact || mxWorld.m[5] <= 0.35f) {
        st.ndSlip = 0; st.Fy = 0;
    } else {
        sc.surface = hit.surface;
        V3(vHitPos).store(st.unmodifiedContactPoint);
        const float fTest = vHitNorm * vWorldM2;
        if (fTest <= 0.96f) {
            float fTestAcos;
            if (fTest <= -1.0f || fTest >= 1.0f) fTestAcos = 0; else fTestAcos = m_acosf(fTest);
            const float fAngle = fTestAcos - m_acosf(0.96f);
            const V3 vAxis((vWorldM2.z * vHitNorm.y) - (vWorldM2.y * vHitNorm.z), (vWorldM2.x * vHitNorm.z) - (vWorldM2.z * vHitNorm.x),
                           (vWorldM2.y * vHitNorm.x) - (vWorldM2.x * vHitNorm.y));
            const M44 mh = axisAngle(vAxis.get_norm(), fAngle);
            vHitNorm = V3((((mh.m[0] * vHitNorm.x) + (mh.m[4] * vHitNorm.y)) + (mh.m[8] * vHitNorm.z)) + mh.m[12],
                          (((mh.m[1] * vHitNorm.x) + (mh.m[5] * vHitNorm.y)) + (mh.m[9] * vHitNorm.z)) + mh.m[13],
                          (((mh.m[2] * vHitNorm.x) + (mh.m[6] * vHitNorm.y)) + (mh.m[10] * vHitNorm.z)) + mh.m[14]);
        } else {
            const V3 vHitOff = vHitPos - worldPosition;
            const float fDot = vHitNorm * vHitOff;
            vHitPos = (vHitNorm * fDot) + worldPosition;
        }
        V3 contactPoint = vHitPos;
        const V3 contactNormal = vHitNorm;
        if (surf) {
            const float fSinHeight = surf->sinHeight;
            if (fSinHeight != 0.0f) {
                const float fSinLength = surf->sinLength;
                contactPoint.y -= (((m_sinf(fSinLength * contactPoint.x) * m_cosf(fSinLength * contactPoint.z)) + 1.0f) * fSinHeight);
            }
            if (surf->granularity != 0.0f) {
                const float v1[3] = {1.0f, 5.8f, 11.4f};
                const float v2[3] = {0.005f, 0.005f, 0.01f};
                const float cx = contactPoint.x, cz = contactPoint.z;
                float cy = contactPoint.y;
                for (int id = 0; id < 3; ++id) { const float v = v1[id]; cy = cy + ((((m_sinf(v * cx) * m_cosf(v * cz)) + 1.0f) * v2[id]) * -0.6f); }
                contactPoint.y = cy;
            }
        }
        contactPoint.store(st.contactPoint);
        contactNormal.store(st.contactNormal);

        // ---- addGroundContact ----
        {
            const V3 vOffset = worldPosition - contactPoint;
            const float fDistToGround = vOffset.len();
            sc.distToGround = fDistToGround;
            float fRadius;
            if (tp.radiusRaiseK == 0.0f) fRadius = tp.radius;
            else fRadius = (fabsf(st.angularVelocity) * tp.radiusRaiseK) + tp.radius;
            if (st.inflation < 1.0f) fRadius = ((fRadius - tp.rimRadius) * st.inflation) + tp.rimRadius;
            sc.liveRadius = fRadius;
            st.effectiveRadius = fRadius;
            if (fDistToGround > fRadius) {
                st.loadedRadius = fRadius; sc.depth = 0; st.load = 0; st.Fy = 0; st.Fx = 0; st.Mz = 0; st.ndSlip = 0;
            } else {
                const float fDepth = fRadius - fDistToGround;
                const float fLoadedRadius = fRadius - fDepth;
                sc.depth = fDepth;
                st.loadedRadius = fLoadedRadius;
                float fMaybePressure;
                if (fLoadedRadius <= tp.rimRadius) fMaybePressure = 200000.0f;
                else {
                    fMaybePressure = ((st.pressureDynamic - tp.pressureRef) * tp.pressureSpringGain) + tp.k;
                    if (fMaybePressure < 0.0f) fMaybePressure = 0;
                }
                const V3 vHubVel = hubPointVelocity(w, su, contactPoint);
                const float fLoad = -((vHubVel * contactNormal) * tp.d) + (fDepth * fMaybePressure);
                st.load = fLoad;
                hubAddForceAtPos(w, su, contactNormal * fLoad, contactPoint);
                if (st.load < 0.0f) st.load = 0;
            }
        }

        // ---- addTyreForcesV10 ----
        {
            const V3 pos = contactPoint, normal = contactNormal;
            V3 vNegM3(worldRotation.m[8], worldRotation.m[9], worldRotation.m[10]);
            vNegM3 = vNegM3 * -1.0f;
            V3 roadHeading = vNegM3 - normal * (vNegM3 * normal);
            roadHeading.norm();
            const V3 vM1(worldRotation.m[0], worldRotation.m[1], worldRotation.m[2]);
            V3 roadRight = vM1 - normal * (vM1 * normal);
            roadRight.norm();
            const V3 hubAngVel(w.bodies[su.hubBody].avel);
            const V3 hubPointVel = hubPointVelocity(w, su, pos);
            sc.slidingVelocityY = hubPointVel * roadRight;
            sc.roadVelocityX = -(hubPointVel * roadHeading);
            float fSlipAngleTmp = calcSlipAngleRAD(sc.slidingVelocityY, sc.roadVelocityX);
            const float fTmp = (hubAngVel * vM1) + st.angularVelocity;
            sc.slidingVelocityX = (fTmp * st.effectiveRadius) - sc.roadVelocityX;
            const float fRoadVelocityXAbs = fabsf(sc.roadVelocityX);
            float fSlipRatioTmp = ((fRoadVelocityXAbs == 0.0f) ? 0.0f : (sc.slidingVelocityX / fRoadVelocityXAbs));
            st.camberRAD = calcCamberRAD(contactNormal, worldRotation);
            sc.totalHubVelocity = sqrtf((sc.roadVelocityX * sc.roadVelocityX) + (sc.slidingVelocityY * sc.slidingVelocityY));
            const float fNdSlip = tclamp(st.ndSlip, 0.0f, 1.0f);
            const float fLoadDivFz0 = st.load / tp.modelFz0;
            const float fRelaxLen = tp.relaxationLength;
            const float fRelax1 = (((fLoadDivFz0 * fRelaxLen) - fRelaxLen) * 0.3f) + fRelaxLen;
            const float fRelax2 = ((fRelaxLen - (fRelax1 * 2.0f)) * fNdSlip) + (fRelax1 * 2.0f);
            if (sc.totalHubVelocity < 1.0f) {
                fSlipRatioTmp = sc.slidingVelocityX * 0.5f;
                fSlipRatioTmp = tclamp(fSlipRatioTmp, -1.0f, 1.0f);
                fSlipAngleTmp = sc.slidingVelocityY * -5.5f;
                fSlipAngleTmp = tclamp(fSlipAngleTmp, -1.0f, 1.0f);
            }
            const float fSlipRatio = st.slipRatio;
            const float fSlipRatioDelta = fSlipRatioTmp - fSlipRatio;
            float fNewSlipRatio = fSlipRatioTmp;
            if (fRelax2 != 0.0f) {
                const float sc2 = (sc.totalHubVelocity * dt) / fRelax2;
                if (sc2 <= 1.0f) {
                    if (sc2 < 0.04f) fNewSlipRatio = (0.04f * fSlipRatioDelta) + fSlipRatio;
                    else fNewSlipRatio = (sc2 * fSlipRatioDelta) + fSlipRatio;
                }
            }
            const float fSlipAngle = st.slipAngleRAD;
            const float fSlipAngleDelta = fSlipAngleTmp - fSlipAngle;
            float fNewSlipAngle = fSlipAngleTmp;
            if (fRelax2 != 0.0f) {
                const float sc2 = (sc.totalHubVelocity * dt) / fRelax2;
                if (sc2 <= 1.0f) {
                    if (sc2 < 0.04f) fNewSlipAngle = (0.04f * fSlipAngleDelta) + fSlipAngle;
                    else fNewSlipAngle = (sc2 * fSlipAngleDelta) + fSlipAngle;
                }
            }
            st.slipAngleRAD = fNewSlipAngle;
            st.slipRatio = fNewSlipRatio;
            if (st.load <= 0.0f) { st.slipAngleRAD = 0; st.slipRatio = 0; }
            // getCorrectedD(1.0, &wearMult)
            float fCorrectedD = (1.0f * st.thermalMultD) / ((fabsf(st.pressureDynamic - tp.idealPressure) * tp.pressureGainD) + 1.0f);
            if (tp.wearCurve.n) { const float wm = curve(tp.wearCurve, (float)st.virtualKM); fCorrectedD *= wm; sc.wearMult = wm; }
            TMI tmi;
            tmi.load = st.load; tmi.slipAngleRAD = st.slipAngleRAD; tmi.slipRatio = st.slipRatio; tmi.camberRAD = st.camberRAD;
            tmi.speed = sc.totalHubVelocity;
            tmi.u = (fCorrectedD * surf->gripMod) * c.T->h->dynamicGripLevel;
            tmi.cpLength = calcContactPatchLength(sc.liveRadius, sc.depth);
            tmi.grain = 0.0f; tmi.blister = 0.0f;
            tmi.pressureRatio = (st.pressureDynamic / tp.idealPressure) - 1.0f;
            tmi.useSimpleModel = false;
            const TMO tmo = sctmSolve(tp, tmi);
            st.Fy = tmo.Fy * 1.0f;
            st.Fx = -tmo.Fx;
            sc.Dy = tmo.Dy; sc.Dx = tmo.Dx;
            float fHubSpeed = st.effectiveRadius * st.angularVelocity;
            // stepDirtyLevel(dt, |hubSpeed|)
            {
                const float hs = fabsf(fHubSpeed);
                if (st.dirtyLevel < 5.0f) st.dirtyLevel += (((hs * surf->dirtAdditiveK) * 0.03f) * dt);
                if (surf->dirtAdditiveK == 0.0f) {
                    if (st.dirtyLevel > 0.0f) st.dirtyLevel -= ((hs * 0.015f) * dt);
                    if (st.dirtyLevel < 0.0f) st.dirtyLevel = 0;
                }
                const float fM = tmax(0.8f, (1.0f - tclamp(st.dirtyLevel * 0.05f, 0.0f, 1.0f)));
                st.Fy *= fM; st.Fx *= fM; st.Mz *= fM;
            }
            // stepPuncture
            if (P.mechanicalDamageRate > 0.0f) {
                float imo[3];
                for (int s = 0; s < 3; ++s) { float sum = 0; for (int j = 0; j < 12; ++j) sum += st.T[j + s * 12]; imo[s] = sum / 12.0f; }
                const float fTemp = tp.explosionTemperature;
                if (imo[0] > fTemp || imo[1] > fTemp || imo[2] > fTemp) st.inflation = 0;
            }
            st.Mz = tmo.Mz;
            V3 vForce = (roadHeading * st.Fx) + (roadRight * st.Fy);
            if (!(std::isfinite(vForce.x) && std::isfinite(vForce.y) && std::isfinite(vForce.z))) vForce = V3(0, 0, 0);
            hubAddForceAtPos(w, su, vForce, pos);
            st.localMX = -(st.loadedRadius * st.Fx);
            hubAddTorque(w, su, normal * tmo.Mz);
            const float fAngularVelocityAbs = fabsf(st.angularVelocity);
            if (fAngularVelocityAbs > 1.0f) {
                fHubSpeed = st.effectiveRadius * st.angularVelocity;
                const float fHubSpeedSign = signf_(fHubSpeed);
                const float fPressureDynamic = st.pressureDynamic;
                float fPressureUnk = (((tp.idealPressure / fPressureDynamic) - 1.0f) * tp.pressureRRGain) + 1.0f;
                if (fPressureDynamic <= 0.0f) fPressureUnk = 0;
                float fRrUnk = ((((fHubSpeed * fHubSpeed) * tp.rr1) + tp.rr0) * fHubSpeedSign) * fPressureUnk;
                if (fAngularVelocityAbs > 20.0f) {
                    const float fNdSlipNorm = tclamp(st.ndSlip, 0.0f, 1.0f);
                    const float fRrSlipUnk = fPressureUnk * tp.rr_slip;
                    const float fSlipUnk = fNdSlipNorm * fRrSlipUnk;
                    fRrUnk = fRrUnk * ((fSlipUnk * 0.001f) + 1.0f);
                }
                sc.rollingResistence = -(((st.load * 0.001f) * fRrUnk) * st.effectiveRadius);
            }
            {
                const float svx = sc.slidingVelocityX, svy = sc.slidingVelocityY;
                const float fSlidingVelocity = sqrtf(svx * svx + svy * svy);
                float fLoadVKM = 1.0f;
                // useLoadForVKM: [VIRTUALKM] USE_LOAD; tyreConsumptionRate = 0 keeps virtualKM at 0 either way
                st.virtualKM += (((fSlidingVelocity * dt) * P.tyreConsumptionRate) * fLoadVKM) * 0.001f;
            }
            const float fStaticDy = sctmStaticDY(tp, st.load);
            st.ndSlip = tmo.ndSlip;
            st.D = fStaticDy;
        }
        if (surf && surf->damping > 0.0f) {
            Body& body = w.bodies[PDB_BODY_CHASSIS];
            const V3 vBodyVel = getVelocity(body);
            const V3 vForce = vBodyVel * -(body.mass * surf->damping);
            const V3 z(0, 0, 0);
            body.addForceAtRelPos(&vForce.x, &z.x);
        }
    }

    // LB_COMPUTE_TORQ
    const float fHandBrakeTorque = sc.handBrakeTorque;
    float fBrakeTorque = sc.brakeTorque * 1.0f;
    if (fBrakeTorque <= fHandBrakeTorque) fBrakeTorque = fHandBrakeTorque;
    const float fAngularVelocitySign = signf_(st.angularVelocity);
    float fTorq = sc.rollingResistence - ((fAngularVelocitySign * fBrakeTorque) + st.localMX);
    if (!std::isfinite(fTorq)) fTorq = 0;
    float fFeedbackTorque = fTorq + 0.0f;
    if (!std::isfinite(fFeedbackTorque)) fFeedbackTorque = 0;
    sc.feedbackTorque = fFeedbackTorque;
    if (tp.driven) {
        updateLockedState(tp, st, sc);
        const float fS0 = signf_(st.oldAngularVelocity);
        const float fS1 = signf_(st.angularVelocity);
        if (fS0 != fS1 && sc.totalHubVelocity < 1.0f) st.isLocked = 1;
        st.oldAngularVelocity = st.angularVelocity;
    } else {
        // updateAngularSpeed
        updateLockedState(tp, st, sc);
        const float fAngVel = st.angularVelocity + ((sc.feedbackTorque / tp.angularInertia) * dt);
        if (signf_(fAngVel) != signf_(st.oldAngularVelocity)) st.isLocked = 1;
        st.oldAngularVelocity = fAngVel;
        st.angularVelocity = st.isLocked ? 0.0f : fAngVel;
        if (fabsf(st.angularVelocity) < 1.0f) st.angularVelocity *= 0.9f;
    }
    if (sc.totalHubVelocity < 10.0f) sc.slipFactor = fabsf(sc.totalHubVelocity * 0.1f) * sc.slipFactor;

    // stepThermalModel
    {
        float fThermalInput = (sqrtf((sc.slidingVelocityX * sc.slidingVelocityX) + (sc.slidingVelocityY * sc.slidingVelocityY)) * ((st.D * st.load) * tp.thermalFrictionK)) * c.T->h->dynamicGripLevel;
        if (sc.surface >= 0) fThermalInput *= c.T->surfaces[sc.surface].gripMod;
        sc.thermalInput = fThermalInput;
        if (std::isfinite(fThermalInput)) {
            const float fPressureDynamic = st.pressureDynamic;
            const float fIdealPressure = tp.idealPressure;
            float fThermalRollingK = tp.thermalRollingK;
            const float fScale = (((fIdealPressure / fPressureDynamic) - 1.0f) * tp.pressureRRGain) + 1.0f;
            if (fPressureDynamic >= 0.0) fThermalRollingK *= fScale;
            const int iVer = tp.version;
            if (iVer < 5) sc.thermalInput += (((fThermalRollingK * st.angularVelocity) * st.load) * 0.001f);
            if (iVer >= 6) sc.thermalInput += ((((fScale * tp.thermalRollingSurfaceK) * st.angularVelocity) * st.load) * 0.001f);
            thermalAddInput(P, tp, st, inputT, st.camberRAD, (fPressureDynamic / fIdealPressure) - 1.0f, sc.thermalInput);
            if (iVer >= 5) coreTInput += (((fThermalRollingK * st.angularVelocity) * st.load) * 0.001f);
            thermalStep(P, tp, st, inputT, coreTInput, dt, st.angularVelocity, st.camberRAD, c.S.speed);
            st.inputT0 = 0;
        }
    }
    st.pressureDynamic = ((st.coreTemp - 26.0f) * tp.pressureTemperatureGain) + tp.pressureStatic;
    // stepGrainBlister: tyreConsumptionRate == 0 -> grain = blister = 0 (Tyre.cpp:917-921)
    // stepFlatSpot
    if (fabsf(st.angularVelocity) <= 0.3f || st.slipRatio < -0.98f) {
        if (sc.surface >= 0 && sc.totalHubVelocity > 3.0f) {
            const float fDamage = P.mechanicalDamageRate;
            if (fDamage != 0.0f) {
                const float fGrip = c.T->surfaces[sc.surface].gripMod;
                if (fGrip >= 0.95f) {
                    st.flatSpot += (((sc.totalHubVelocity * tp.flatSpotK) * st.load) * fGrip) * 0.00001f * dt * fDamage * tp.softnessIndex;
                    if (st.flatSpot > 1.0f) st.flatSpot = 1.0f;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Car::getGroundWindVector / getPointGroundHeight (Car.cpp:1367-1405), Wing::step (Wing.cpp:71-204)
// ------------------------------------------------------------------------------------------------
struct Plane { V3 normal; float d; Plane(const V3& p1, const V3& p2, const V3& p3) { const V3 a = p3 - p1, b = p2 - p1; normal = a.cross(b); d = -(p1 * normal); } };

static inline float kmhOfSpeed(float ms) { return ms * 3.6f; }   // Speed::kmh
// AeroMap::addDrag / addLift (AeroMap.cpp:99-139)
static void aeroDataStep(Car& c) {
    const pdb_car_params& P = *c.P;
    Body& body = c.w.bodies[PDB_BODY_CHASSIS];
    const V3 lv = w2lN(body, getVelocity(body));
    {
        float fDot = lv.sqlen();
        if (fDot != 0.0f) {
            V3 vNorm = lv / sqrtf(fDot);
            const float dynamicCD = (((fabsf(vNorm.x) * P.aeroCD) * P.aeroCDX) + P.aeroCD) + ((fabsf(vNorm.y) * P.aeroCD) * P.aeroCDY);
            const float fDrag = ((dynamicCD * fDot) * c.airDensityNow) * P.aeroReferenceArea;
            const V3 f = vNorm * -(fDrag * 0.5f);
            const float zero[3] = {0, 0, 0};
            body.addRelForceAtRelPos(&f.x, zero);
            const V3 vAngVel(body.avel);
            fDot = vAngVel.sqlen();
            if (fDot != 0.0f) {
                vNorm = vAngVel / sqrtf(fDot);
                const V3 t = vNorm * -(fDot * P.aeroCDA);
                body.addRelTorque(&t.x);
            }
        }
    }
    {
        const float fZZ = lv.z * lv.z;
        if (fZZ != 0.0f) {
            const float fLift = (((fZZ * P.aeroCL) * c.airDensityNow) * P.aeroReferenceArea) * 0.5f;
            const float fFrontLift = fLift * P.aeroFrontShare;
            const V3 ff(0, -fFrontLift, 0);
            body.addRelForceAtRelPos(&ff.x, P.susp[0].basePosition);
            const float fRearLift = fLift * (1.0f - P.aeroFrontShare);
            const V3 fr(0, -fRearLift, 0);
            body.addRelForceAtRelPos(&fr.x, P.susp[2].basePosition);
        }
    }
}

static void wingStep(Car& c, int wi) {
    const pdb_car_params& P = *c.P;
    const pdb_wing& wg = P.wings[wi];
    WingScratch& ws = c.ws[wi];
    Body& body = c.w.bodies[PDB_BODY_CHASSIS];
    V3 vGroundWind;
    {
        const Plane ground(V3(c.S.tyre[0].unmodifiedContactPoint), V3(c.S.tyre[1].unmodifiedContactPoint), V3(c.S.tyre[2].unmodifiedContactPoint));
        const V3 wind(0, 0, 0);
        const float dot = (wind * ground.normal);
        vGroundWind = (wind - (ground.normal * dot)) * 0.44f;
    }
    const V3 pos(wg.position);
    const V3 vWorldVel = localPointVel(body, pos);
    const V3 vLocalVel = w2lN(body, vWorldVel + vGroundWind);
    const V3 vWingWorld = l2w(body, pos);
    {
        const V3 c0(c.S.tyre[0].contactPoint), c1(c.S.tyre[1].contactPoint), c2(c.S.tyre[2].contactPoint), c3(c.S.tyre[3].contactPoint);
        const Plane pl1(c0, c1, c2), pl2(c0, c1, c3);
        const V3 axis(0, -1, 0);
        const float dot1 = pl1.normal * axis, dot2 = pl2.normal * axis;
        float y1 = 0, y2 = 0;
        const V3& pt = vWingWorld;
        if (dot1 != 0.0f) y1 = pt.y + ((pt * pl1.normal + pl1.d) / dot1);
        if (dot2 != 0.0f) y2 = pt.y + ((pt * pl2.normal + pl2.d) / dot2);
        ws.groundHeight = ((pt.y - y2) + (pt.y - y1)) * 0.5f;
    }
    // Wing::stepDynamicControllers (Wing.cpp:105-124) + WingDynamicController::step (WingDynamicController.cpp:64-75)
    float angle = wg.angle;
    for (int j = 0; j < P.numWingCtrl; ++j) {
        const pdb_wing_ctrl& wc = P.wingCtrl[j];
        if (wc.wing != wi) continue;
        float in = 0.0f;
        switch (wc.input) {
            case 1: in = c.controls.brake; break;
            case 2: in = c.controls.gas; break;
            case 3: in = c.accG[0]; break;
            case 4: in = c.accG[2]; break;
            case 5: in = c.controls.steer; break;
            case 6: in = kmhOfSpeed(c.S.speed); break;
            case 7: in = c.S.suspTravel[2]; break;
            case 8: in = c.S.suspTravel[3]; break;
        }
        float fAngle = curve(wc.lut, in);
        const float out = c.S.wingCtrlOut[j];
        if (fabsf(fAngle - out) >= 0.001f) fAngle = out + (fAngle - out) * tclamp(wc.filter * 0.003f, 0.0f, 1.0f);
        c.S.wingCtrlOut[j] = fAngle;
        if (wc.combinator == 1) angle += fAngle; else if (wc.combinator == 2) angle *= fAngle;
        angle = tclamp(angle, wc.downLimit, wc.upLimit);
    }
    if (vLocalVel.z == 0.0f) { ws.aoa = 0; ws.yawAngle = 0; ws.cd = 0; ws.cl = 0; return; }
    ws.aoa = m_atanf((1.0f / vLocalVel.z) * vLocalVel.y) * 57.29578f;
    ws.yawAngle = m_atanf((1.0f / vLocalVel.z) * vLocalVel.x) * 57.29578f;
    const V3& lv = vLocalVel;
    {   // addDrag
        const float off = wg.isVertical ? ws.yawAngle : ws.aoa;
        ws.cd = curve(wg.lutAOA_CD, (1.0f * angle) + off) * wg.cdGain;
        if (wg.lutGH_CD.n) ws.cd *= curve(wg.lutGH_CD, ws.groundHeight);   // Wing.cpp:134-139
        const float fDot = lv.sqlen();
        const float fDrag = (((fDot * ws.cd) * c.airDensityNow) * wg.area) * 0.5f;
        ws.dragKG = fDrag * 0.10197838f;
        if (fDot != 0.0f) { const V3 f = lv.get_norm() * -fDrag; body.addRelForceAtRelPos(&f.x, &pos.x); }
    }
    {   // addLift
        float off, fAxis;
        if (wg.isVertical) { off = ws.yawAngle; fAxis = lv.x; } else { off = ws.aoa; fAxis = lv.y; }
        ws.cl = curve(wg.lutAOA_CL, (1.0f * angle) + off) * wg.clGain;
        if (lv.z < 0.0f) ws.cl = 0;
        if (!wg.isVertical && wg.yawGain != 0.0f) {
            const float v8 = (m_sinf(fabsf(ws.yawAngle) * 0.017453f) * wg.yawGain) + 1.0f;
            ws.cl *= tclamp(v8, 0.0f, 1.0f);
        }
        if (wg.lutGH_CL.n) ws.cl *= curve(wg.lutGH_CL, ws.groundHeight);   // Wing.cpp:181-186
        const float fDot = (fAxis * fAxis) + (lv.z * lv.z);
        const float fLift = (((fDot * ws.cl) * c.airDensityNow) * wg.area) * 0.5f;
        ws.liftKG = fLift * 0.10197838f;
        if (fDot != 0.0f) {
            const V3 vNorm = lv.get_norm();
            const V3 vOut = wg.isVertical ? V3(-vNorm.z, 0, vNorm.x) : V3(0, vNorm.z, -vNorm.y);
            const V3 vForce = vOut * -fLift;
            body.addRelForceAtRelPos(&vForce.x, &pos.x);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Drivetrain / Engine / assists
// ------------------------------------------------------------------------------------------------
static float engineRpm(const Car& c) { return (float)((c.S.engineVel * 0.15915507) * 60.0); }        // Drivetrain::getEngineRPM
static float carEngineRpm(const Car& c) { return ((float)c.S.engineVel * 0.15915507f * 60.0f); }     // Car::getEngineRpm (Car.cpp:1431-1434)
static int limiterRpm(const pdb_car_params& P) { return (int)(P.engLimiter * P.limiterMultiplier); }
static float kmh(float ms) { return ms * 3.6f; }

// AutoClutch::onGearRequest + AutoBlip handler (AutoClutch.cpp:202-221, AutoBlip.cpp:36-48)
static void fireGearRequest(Car& c, int request /*1 up, 2 down*/) {
    const pdb_car_params& P = *c.P;
    if (P.acUseOnChange) {
        if (request == 2 && c.S.acClutchValueSignal > 0.01f && P.downshiftProfile.n == 4) { c.S.acSeqActive = 2; c.S.acSeqCurrentTime = 0; c.S.acSeqIsDone = 0; }
        if (request == 1 && c.S.acClutchValueSignal > 0.01f && P.upshiftProfile.n == 4) { c.S.acSeqActive = 1; c.S.acSeqCurrentTime = 0; c.S.acSeqIsDone = 0; }
    }
    if (request == 2 && c.controls.clutch > 0.1f) c.S.blipStartTime = (c.S.physicsTime * 1000.0);
}
// Drivetrain::gearUp / gearDown (Drivetrain.cpp:209-279)
static bool gearUp(Car& c) {
    const pdb_car_params& P = *c.P;
    const int req = c.S.currentGear + 1;
    if (req >= P.numGears) return false;
    if (c.S.gearReqRequest != 0) return false;
    c.S.gearReqRequest = 1; c.S.gearReqTimeAccumulator = 0; c.S.gearReqTimeout = P.gearUpTime; c.S.gearReqRequestedGear = req;
    fireGearRequest(c, 1);
    if (P.autoCutOffTime != 0.0f) c.S.cutOff = P.autoCutOffTime;
    c.S.currentGear = 1;
    return true;
}
static bool gearDown(Car& c) {
    const pdb_car_params& P = *c.P;
    const int cur = c.S.currentGear;
    const int req = cur - 1;
    if (cur <= 0) return false;
    if (c.S.gearReqRequest != 0) return false;
    c.S.gearReqRequest = 2; c.S.gearReqTimeAccumulator = 0; c.S.gearReqTimeout = P.gearDnTime; c.S.gearReqRequestedGear = req;
    fireGearRequest(c, 2);
    c.S.currentGear = 1;
    return true;
}

// AutoClutch::step / stepSequence (AutoClutch.cpp:91-200)
static void autoClutchStep(Car& c, float dt) {
    const pdb_car_params& P = *c.P;
    pdb_dyn_state& S = c.S;
    if (!S.acSeqIsDone) {
        const bool moving = kmh(S.speed) >= 5.0f;
        if (moving) {
            const pdb_curve& cv = (S.acSeqActive == 1) ? P.upshiftProfile : P.downshiftProfile;
            const float sig = curve(cv, S.acSeqCurrentTime);
            S.acClutchValueSignal = sig;
            S.acSeqCurrentTime += dt;
            if (S.acSeqCurrentTime > (cv.n ? cv.x[cv.n - 1] : 0.0f)) S.acSeqIsDone = 1;
            c.controls.clutch = tclamp(S.acClutchValueSignal, 0.0f, 1.0f);
            return;
        }
        S.acSeqIsDone = 1;
    }
    if (!P.acUseOnStart) return;
    float fNewClutchInput = 1.0f, fNewSignal = 1.0f;
    const float rpm = engineRpm(c);
    const int gear = S.currentGear;
    const bool stationary = kmh(S.speed) < 5.0f;
    bool toZero = false;
    if ((gear & 0xFFFFFFFD) != 0) {
        if (gear == 1) {
            if (stationary) {
                if (c.controls.gas > 0.2f) S.acClutchValueSignal = 1.0f;
                else toZero = true;
            }
        } else toZero = rpm < P.acRpmMin;
    } else {
        if (rpm >= P.acRpmMin && rpm <= P.acRpmMax) { fNewSignal = (rpm - P.acRpmMin) / (P.acRpmMax - P.acRpmMin); S.acClutchValueSignal = fNewSignal; }
        if (rpm > P.acRpmMax) fNewSignal = 1.0f;
        toZero = rpm < P.acRpmMin;
    }
    if (toZero) { fNewSignal = 0.0f; S.acClutchValueSignal = 0.0f; }
    const float cur = S.acClutchValueSignal;
    const float sd = dt * P.acClutchSpeed;
    if (fabsf(fNewSignal - cur) >= sd) { if (fNewSignal <= cur) S.acClutchValueSignal = cur - sd; else S.acClutchValueSignal = sd + cur; }
    else S.acClutchValueSignal = fNewSignal;
    if (S.acClutchValueSignal <= 1.0f) { if (S.acClutchValueSignal >= 0.0f) fNewClutchInput = S.acClutchValueSignal; else fNewClutchInput = 0.0f; }
    c.controls.clutch = fNewClutchInput;
}

// Engine::step (Engine.cpp:193-342) -- no turbos, no coast generators, no overlap
// DynamicController::eval / getInput (Car/DynamicController.cpp:117-166,168-255; the inputs the loader admits)
static float dynCtrlEval(Car& c, const pdb_dyn_ctrl& dc) {
    const pdb_car_params& P = *c.P;
    pdb_dyn_state& S = c.S;
    float fRes = 0;
    for (int k = dc.first; k < dc.first + dc.count; ++k) {
        const pdb_ctrl_stage& st = P.ctrlStages[k];
        const float fCurValue = S.ctrlValue[k];
        float fNewValue = 0;
        if (st.input == 9) fNewValue = st.constValue;
        else {
            float in = 0.0f;
            switch (st.input) {
                case 1: in = c.controls.brake; break;
                case 2: in = c.controls.gas; break;
                case 3: in = c.accG[0]; break;
                case 4: in = c.accG[2]; break;
                case 5: in = c.controls.steer; break;
                case 6: in = kmh(S.speed); break;
                case 7: in = (float)(S.currentGear - 1); break;
                case 8: in = engineRpm(c); break;
                default: {   // the tyres' status as it stands when the consumer steps (DynamicController.cpp:191-253)
                    const pdb_tyre_state* t = S.tyre;
                    const int d0 = P.tractionType == 0 ? 2 : 0;   // the driven axle (RWD / FWD)
                    switch (st.input) {
                        case 10: in = tmax(t[d0].slipRatio, t[d0 + 1].slipRatio); break;
                        case 11: in = (t[d0].slipRatio + t[d0 + 1].slipRatio) * 0.5f; break;
                        case 12: in = ((t[0].slipAngleRAD + t[1].slipAngleRAD) * 57.29578f) * 0.5f; break;
                        case 13: in = ((t[2].slipAngleRAD + t[3].slipAngleRAD) * 57.29578f) * 0.5f; break;
                        case 14: in = tmax<float>(fabsf(t[0].slipAngleRAD), fabsf(t[1].slipAngleRAD)) * 57.29578f; break;
                        case 15: in = tmax<float>(fabsf(t[2].slipAngleRAD), fabsf(t[3].slipAngleRAD)) * 57.29578f; break;
                        case 16: in = ((fabsf(t[2].slipAngleRAD) + fabsf(t[3].slipAngleRAD) * 0.5f) - (fabsf(t[0].slipAngleRAD) + fabsf(t[1].slipAngleRAD) * 0.5f)) * 57.29578f; break;
                        case 17: { const float front = (t[1].angularVelocity + t[0].angularVelocity) * 0.5f; in = front != 0.0f ? ((t[3].angularVelocity + t[2].angularVelocity) * 0.5f) / front : 0.0f; } break;
                        case 18: in = P.steerLock * c.controls.steer; break;
                        case 19: in = c.finalSteerAngleSignal; break;
                        case 20: in = t[0].load / (t[0].load + t[1].load); break;
                        case 21: in = t[1].load / (float)(t[1].load + t[0].load); break;
                        case 22: in = (S.suspTravel[2] + S.suspTravel[3]) * 0.5f * 1000.0f; break;
                        case 23: in = S.suspTravel[2] * 1000.0f; break;
                        case 24: in = S.suspTravel[3] * 1000.0f; break;
                    }
                } break;
            }
            fNewValue = curve(st.lut, in);
        }
        if (fabsf(fNewValue - fCurValue) >= 0.001f) {
            const float fFilter = tclamp(st.filter * 0.003f, 0.0f, 1.0f);
            fNewValue = ((fNewValue - fCurValue) * fFilter) + fCurValue;
        }
        S.ctrlValue[k] = fNewValue;
        if (st.combinator == 1) fRes += fNewValue; else fRes *= fNewValue;
        if (st.downLimit != 0.0f || st.upLimit != 0.0f) fRes = tclamp(fRes, st.downLimit, st.upLimit);
    }
    return fRes;
}

static void engineStep(Car& c, float gasInput, float rpm) {
    const pdb_car_params& P = *c.P;
    pdb_dyn_state& S = c.S;
    // Engine::getThrottleResponseGas (Engine.cpp:344-366)
    float gas;
    if (P.throttleCurve.n && P.throttleCurveMax.n) {
        const float fTrc = tclamp(curve(P.throttleCurve, gasInput * 100.0f) * 0.01f, 0.0f, 1.0f);
        const float fTrcMax = tclamp(curve(P.throttleCurveMax, gasInput * 100.0f) * 0.01f, 0.0f, 1.0f);
        const float fTrcScale = tclamp(rpm / P.throttleMaxRef, 0.0f, 1.0f);
        gas = ((fTrcMax - fTrc) * fTrcScale) + fTrc;
    } else if (P.throttleCurve.n) gas = tclamp(curve(P.throttleCurve, gasInput * 100.0f) * 0.01f, 0.0f, 1.0f);
    else gas = gasInput;
    if (P.gasCoastOffset > 0.0f) {   // [COAST_SETTINGS] (Engine.cpp:198-207)
        float fGas1 = (rpm - (float)P.engMinimum) / (float)P.coastEntryRpm;
        fGas1 = tclamp(fGas1, 0.0f, 1.0f);
        float fGas2 = ((1.0f - (P.gasCoastOffset * fGas1)) * gas) + (P.gasCoastOffset * fGas1);
        fGas2 = tclamp(fGas2, 0.0f, 1.0f);
        gas = fGas2;
    }
    const int iLimiter = P.engLimiter;
    if (iLimiter && (iLimiter * P.limiterMultiplier) < rpm) S.limiterOn = P.engLimiterCycles;
    if (S.limiterOn > 0) { gas = 0; S.limiterOn--; }
    if (S.lifeLeft <= 0.0f) S.fuelPressure = 0;
    const float fGas = gas * 1.0f;
    c.gasUsage = fGas;
    S.gasUsage = fGas;
    float fPower = curve(P.powerCurve, rpm);
    float fCoastTorq = 0;
    if (P.engCoast1 != 0.0f) fCoastTorq = (rpm - (float)P.engMinimum) * P.engCoast1;
    // stepTurbos (Engine.cpp:368-384) -> Turbo::step (Turbo.cpp:11-38) with the fixed 0.003 s step of the reference
    float turboBoost = 0;
    for (int t = 0; t < P.numTurbos; ++t) {
        pdb_turbo tb = P.turbos[t];
        // Engine::stepTurbos' controllers (Engine.cpp:370-376): the turbo's maxBoost / wastegate of this tick
        if (P.ctrlTurboBoost[t].count) tb.maxBoost = dynCtrlEval(c, P.ctrlTurboBoost[t]);
        if (P.ctrlWastegate[t].count) tb.wastegate = dynCtrlEval(c, P.ctrlWastegate[t]);
        float rot = S.turboRotation[t];
        float fNewRotation = 0, fLag;
        if (rpm > 0.0f && fGas > 0.0f) fNewRotation = m_powf(tclamp(((fGas * rpm) / tb.rpmRef), 0.0f, 1.0f), tb.gamma);
        if (fNewRotation <= rot) fLag = tclamp((0.003f * tb.lagDN), 0.0f, 1.0f);
        else fLag = tclamp((0.003f * tb.lagUP), 0.0f, 1.0f);
        rot += ((fNewRotation - rot) * fLag);
        if (tb.wastegate != 0.0f) {
            const float fUserWG = tb.wastegate * tb.userSetting;
            if ((tb.maxBoost * rot) > fUserWG) rot = fUserWG / tb.maxBoost;
        }
        S.turboRotation[t] = rot;
        turboBoost += ((tb.maxBoost * rot) * S.fuelPressure);
    }
    c.turboBoost = turboBoost;
    if (turboBoost != 0.0f) fPower *= (turboBoost + 1.0f);
    if (P.engCoast2 != 0.0f) { const float d = rpm - (float)P.engMinimum; fCoastTorq -= (((d * d) * P.engCoast2) * signf_(rpm)); }
    fCoastTorq += (float)0.0;
    if (rpm <= (float)P.engMinimum) fCoastTorq = 0;
    if (P.turboBoostDamageThreshold != 0.0f && turboBoost > P.turboBoostDamageThreshold)
        S.lifeLeft -= ((((turboBoost - P.turboBoostDamageThreshold) * P.turboBoostDamageK) * 0.003f) * P.mechanicalDamageRate);
    if (P.rpmDamageThreshold != 0.0f && rpm > P.rpmDamageThreshold)
        S.lifeLeft -= ((((rpm - P.rpmDamageThreshold) * P.rpmDamageK) * 0.003f) * P.mechanicalDamageRate);
    const float fAirAmount = P.airDensity * 0.82630974f;
    const float fOutTorq = ((((fPower - fCoastTorq) * fGas) + fCoastTorq) * fAirAmount);
    c.engOutTorque = fOutTorq;
    if (S.fuelPressure > 0.0f) {
        if (rpm >= (float)P.engMinimum) {
            if (P.overlapGain != 0.0f) {   // [OVERLAP] (Engine.cpp:300-307)
                const float fOverlap = m_sinf((float)S.physicsTime * 0.001f * P.overlapFreq * rpm * 0.0003333333333333333f) * 0.5f - 0.5f;
                c.engOutTorque = (fOverlap * fabsf(rpm - P.overlapIdealRPM) * P.overlapGain) + fOutTorq;
            }
        } else c.engOutTorque = tmax(15.0f, fOutTorq);
    }
    if (S.fuelPressure < 1.0f) c.engOutTorque = (c.engOutTorque - rpm * -0.01f) * S.fuelPressure + rpm * -0.01f;
}

// Drivetrain::step / step2WD / reallignSpeeds / accelerateDrivetrainBlock / getInertia* (Drivetrain.cpp:281-725)
static void accelBlock(Car& c, double acc) { c.S.driveVel += acc; c.S.outShaftRVel += acc; c.S.outShaftLVel += acc; }
static double inertiaFromWheels(const Car& c, double engineInertia) {
    const pdb_car_params& P = *c.P;
    const double r = c.ratio, rr = r * r;
    const double rwd = P.driveInertia + (P.outShaftInertiaL + P.outShaftInertiaR);
    if (r == 0.0) return rwd;
    if (c.S.clutchOpenState) return rwd + (P.clutchInertia * rr);
    return rwd + ((P.clutchInertia + engineInertia) * rr);
}
static double inertiaFromEngine(const Car& c, double engineInertia) {
    const pdb_car_params& P = *c.P;
    const double r = c.ratio;
    if (r == 0.0) return engineInertia;
    const double rwd = P.driveInertia + P.outShaftInertiaL + P.outShaftInertiaR;
    return rwd / (r * r) + P.clutchInertia + engineInertia;
}

static void drivetrainStep(Car& c, float dt) {
    const pdb_car_params& P = *c.P;
    pdb_dyn_state& S = c.S;
    const int tl = (P.tractionType == 0) ? 2 : 0, tr = tl + 1;
    pdb_tyre_state& TL = S.tyre[tl];
    pdb_tyre_state& TR = S.tyre[tr];
    TyreScratch& SL = c.ts[tl];
    TyreScratch& SR = c.ts[tr];
    c.locClutch = m_powf(c.controls.clutch, 1.5f);
    S.locClutch = (float)c.locClutch;   // exact: the value is a float
    c.currentClutchTorque = 0;
    // Drivetrain::stepControllers (Drivetrain.cpp:587-608): ctrl_single_lock.ini drives the differential's preload, the power ramp is then 0
    double diffPreLoad = P.diffPreLoad, diffPowerRamp = P.diffPowerRamp;
    if (P.ctrlDiffLock.count) { diffPreLoad = dynCtrlEval(c, P.ctrlDiffLock); diffPowerRamp = 0; }
    // step2WD
    const int gr = S.gearReqRequest - 1;
    if ((!gr || gr == 1) && (S.gearReqTimeout < S.gearReqTimeAccumulator)) { S.currentGear = S.gearReqRequestedGear; S.gearReqRequest = 0; }
    if (S.gearReqRequest != 0) S.gearReqTimeAccumulator += dt;
    const double curGearRatio = P.gearRatio[S.currentGear];
    c.ratio = P.finalRatio * curGearRatio;
    const double engineInertia = P.engInertia;
    if (S.lastRatio != c.ratio) {
        // reallignSpeeds
        const double fRatio = c.ratio;
        if (fRatio != 0.0) {
            const double fDriveVel = S.driveVel;
            if (c.locClutch <= 0.9f) S.rootVelocity = fDriveVel * fRatio;
            else S.rootVelocity -= (1.0 - engineInertia / inertiaFromEngine(c, engineInertia)) * (S.rootVelocity / fRatio - fDriveVel) * fabs(fRatio);
            accelBlock(c, (S.rootVelocity / fRatio - fDriveVel));
            if (!S.clutchOpenState) S.engineVel = S.rootVelocity;
        }
        S.lastRatio = c.ratio;
    }
    float gasInput = 0;
    if (S.cutOff > 0.0) S.cutOff -= dt; else gasInput = c.controls.gas;
    const float rpm = (float)((S.engineVel * 0.15915507) * 60.0);
    engineStep(c, gasInput, rpm);
    const double outTorque = c.engOutTorque;
    if (c.locClutch < 1.0f) S.clutchOpenState = 1;
    else if (S.engineVel != 0.0) S.clutchOpenState = (fabs(S.rootVelocity / S.engineVel - 1.0) >= 0.1);
    else S.clutchOpenState = (S.rootVelocity != 0.0);
    const double fEngineInertia = engineInertia;
    double fNewEngineInertia = fEngineInertia;
    if (c.ratio != 0.0) {
        const double sum = P.driveInertia + P.outShaftInertiaL + P.outShaftInertiaR;
        fNewEngineInertia = sum / (c.ratio * c.ratio) + P.clutchInertia + fEngineInertia;
    }
    const double fInertiaFromWheels = inertiaFromWheels(c, engineInertia);
    double fDeltaDriveV = 0, fClutchTorq = 0;
    if (!S.clutchOpenState) {
        const double fDeltaRootV = (outTorque / fNewEngineInertia) * dt;
        S.rootVelocity += fDeltaRootV;
        if (c.ratio == 0.0) {
            fDeltaDriveV = (SR.feedbackTorque + SL.feedbackTorque) / fInertiaFromWheels * dt;
            S.driveVel += fDeltaDriveV;
        } else {
            accelBlock(c, fDeltaRootV / c.ratio);
            fDeltaDriveV = (SR.feedbackTorque + SL.feedbackTorque) / fInertiaFromWheels * dt;
            S.rootVelocity += fDeltaDriveV * c.ratio;
            S.driveVel += fDeltaDriveV;
        }
    } else {
        fClutchTorq = -((S.engineVel - S.rootVelocity) / (fabs(S.engineVel - S.rootVelocity) + 4.0) * (c.locClutch * P.clutchMaxTorque));
        c.currentClutchTorque = fClutchTorq;
        if (c.ratio != 0.0) {
            S.engineVel += (fClutchTorq + outTorque) / fEngineInertia * dt;
            const double fDeltaRootV = (-fClutchTorq / (fNewEngineInertia - fEngineInertia)) * dt;
            S.rootVelocity += fDeltaRootV;
            accelBlock(c, fDeltaRootV / c.ratio);
            fDeltaDriveV = (SR.feedbackTorque + SL.feedbackTorque) / fInertiaFromWheels * dt;
            S.rootVelocity += fDeltaDriveV * c.ratio;
            S.driveVel += fDeltaDriveV;
        } else {
            const double v = S.engineVel + outTorque / fEngineInertia * dt;
            S.engineVel = v;
            S.rootVelocity = v;
            fDeltaDriveV = (SR.feedbackTorque + SL.feedbackTorque) / fInertiaFromWheels * dt;
            S.driveVel += fDeltaDriveV;
        }
    }
    S.outShaftLVel += fDeltaDriveV;
    S.outShaftRVel += fDeltaDriveV;
    if (P.diffType == 1) { S.outShaftLVel = S.driveVel; S.outShaftRVel = S.driveVel; }
    else {
        double fOutClutchTorq, fDiffLoad;
        if (fClutchTorq != 0.0) fOutClutchTorq = -fClutchTorq; else fOutClutchTorq = c.locClutch * outTorque;
        if (fOutClutchTorq <= 0.0) fDiffLoad = fabs(c.ratio * P.diffCoastRamp * fOutClutchTorq);
        else fDiffLoad = fabs(c.ratio) * (diffPowerRamp * fOutClutchTorq);
        const double fDiffTotalLoad = fDiffLoad + diffPreLoad;
        if (fabs(S.outShaftLVel - S.driveVel) >= 0.1 || fabs(SR.feedbackTorque - SL.feedbackTorque) > fDiffTotalLoad) {
            const double fUnk1 = -((S.outShaftLVel - S.outShaftRVel) / (fabs(S.outShaftLVel - S.outShaftRVel) + 0.01) * fDiffTotalLoad);
            const double fDeltaV1 = dt * (fUnk1 / P.outShaftInertiaL * 0.5);
            S.outShaftLVel += fDeltaV1;
            S.outShaftRVel -= fDeltaV1;
            const double fDeltaV2 = dt * ((SR.feedbackTorque - SL.feedbackTorque) / P.outShaftInertiaR * 0.5);
            S.outShaftLVel -= fDeltaV2;
            S.outShaftRVel += fDeltaV2;
        } else { S.outShaftLVel = S.driveVel; S.outShaftRVel = S.driveVel; }
    }
    if (TL.isLocked && TR.isLocked) {
        const float fTorqL = (1.0f * SL.brakeTorque) + SL.handBrakeTorque;
        const float fTorqR = (1.0f * SR.brakeTorque) + SR.handBrakeTorque;
        bool bFlag = true;
        if (fabs(c.ratio * outTorque) <= (fTorqL + fTorqR)) { if (S.speed <= 1.0f) bFlag = false; }
        if (bFlag) { TL.isLocked = 0; TR.isLocked = 0; }
        else if (S.clutchOpenState) { S.rootVelocity = 0; S.driveVel = 0; S.outShaftLVel = 0; S.outShaftRVel = 0; }
    }
    if (!S.clutchOpenState) S.engineVel = S.rootVelocity;
    TL.angularVelocity = (float)S.outShaftLVel;
    TR.angularVelocity = (float)S.outShaftRVel;
    if (c.ratio == 0.0) c.totalTorque = fabs(outTorque * c.locClutch);
    else c.totalTorque = fabs((fabs(c.ratio) * (outTorque * c.locClutch)) - (SL.feedbackTorque + SR.feedbackTorque));
    const float fGearTorque = (float)(c.locClutch * outTorque * curGearRatio);
    if (P.suspTypeR == PDB_SUSP_AXLE) {
        const float fAxleTorq = fGearTorque * P.axleTorqueReaction;
        const V3 a(0, 0, fAxleTorq), b(0, 0, -fAxleTorq);
        c.w.bodies[PDB_BODY_CHASSIS].addRelTorque(&a.x);
        c.w.bodies[P.susp[2].hubBody].addRelTorque(&b.x);
    }
}

// ------------------------------------------------------------------------------------------------
// Car::step (Car.cpp:421-553) + stepComponents (:638-681)

// ------------------------------------------------------------------------------------------------
// Body contacts: PhysicsEngineODE::collisionStep / collisionNearCallback / onCollision (PhysicsEngineODE.cpp:228-341) and
// Car::onCollisionCallback (Car.cpp:921-1044).  PARITY UNPINNED: contact generation is ODE's (dCollide box-trimesh and
// trimesh-trimesh), which is not in the reference tree; what follows is this project's own definition of the same
// quantities, chosen so that every result is an OR or a maximum over contacts (independent of any traversal order):
//   * pairs: the chassis' belly box against surfaces whose category meets C_MASK_CAR_BOX (1: TRACK), its hull mesh against
//     surfaces whose category meets C_MASK_CAR_MESH (30: WALL); dynamic-vs-static pairs are collided on odd frames only;
//   * box vs triangle: plane reach test in world space, then the separating-axis test in the box frame; the contact normal is the triangle's, turned towards the
//     box centre, and contacts whose body-local normal.y < 0.9 are dropped (PhysicsEngineODE.cpp:303-312);
//   * hull vs triangle: every edge of one triangle that pierces the other is a contact at the piercing point, with the
//     wall triangle's normal turned towards the chassis origin;
//   * each contact raises collisionFlag; hull contacts feed relative speed -> damage zones / engine blow-up.
//   * contact joints (PhysicsEngineODE::onCollision :283-331): every contact point (oracle/rb/pdcollide.h: box corners / edge
//     midpoints, hull piercing points, each with a depth) is a candidate; the car's PDB_MAX_CONTACTS deepest are kept and stay
//     in the solve until the group is refilled on the next odd frame (:228-243); oracle/rb/pdrb.cpp solves them.
// ------------------------------------------------------------------------------------------------
void Car::collisionStep() {
    const pdb_collider& C = P->collider;
    const int frame = S.simFrame;
    S.simFrame = frame + 1;
    S.damageChanged = 0;
    if (!C.enabled) { S.numContacts = 0; contactSet.clear(); return; }
    if (!(frame & 1)) return;
    contactSet.clear();   // dJointGroupEmpty(contactGroupDynamic)
    contactCandidates = 0;
    const Body& body = w.bodies[PDB_BODY_CHASSIS];
    pdcol::Pose pose;
    memcpy(pose.pos, body.pos, 12); memcpy(pose.R, body.R, 36);
    pdcol::V aLo, aHi;
    pdcol::worldAabb(pose, C.boundsLo, C.boundsHi, aLo, aHi);
    const TrackData& Tk = *T;
    bool flag = false, blow = false;
    float dmg[5] = {0, 0, 0, 0, 0};
    for (int s = 0; s < Tk.h->numSurfaces; ++s) {
        const int cat = Tk.surfaces[s].collisionCategory;
        // collisionNearCallback (PhysicsEngineODE.cpp:258-264): (cat1 & mask2) && (cat2 & mask1), car geoms are category 4,
        // surfaces collide with mask 20, the box with mask 1, the hull with mask 30 (Sim/SimulatorCommon.h:7-13)
        const bool meshPair = (cat & 30) != 0 && C.numTris > 0, boxPair = (cat & 1) != 0 && C.numBoxes != 0;
        if (!meshPair && !boxPair) continue;
        const bool noDamage = (cat == 1 || cat == 16);   // Car.cpp:948-958: bFlag for groups 1 and 16
        for (int t = Tk.surfaces[s].triStart; t < Tk.surfaces[s].triStart + Tk.surfaces[s].triCount; ++t) {
            const pdcol::V p0 = pdcol::ld(Tk.tris + 9 * t), p1 = pdcol::ld(Tk.tris + 9 * t + 3), p2 = pdcol::ld(Tk.tris + 9 * t + 6);
            if (!pdcol::triMeetsAabb(p0, p1, p2, aLo, aHi)) continue;
            if (boxPair) for (int bx = 0; bx < C.numBoxes; ++bx) {   // every box geom of the chassis meets the triangle on its own (CarColliderManager.cpp:17-33: one geom per COLLIDER_n)
                float ny;
                if (pdcol::boxContacts(pose, C.boxCentre[bx], C.boxHalf[bx], p0, p1, p2, ny, [&](const pdcol::V& pw, const pdcol::V& nw, float depth, int item) {
                        contactSet.insert(pw, nw, depth, 1, (unsigned)t * pdcol::ID_STRIDE + (unsigned)item + 16u * (unsigned)bx);
                        ++contactCandidates;
                    }) && ny >= 0.9f) flag = true;   // PhysicsEngineODE.cpp:303-312
            }
            if (meshPair && pdcol::triMeetsBounds(pose, C.boundsLo, C.boundsHi, p0, p1, p2))
                pdcol::hullContacts(pose, C.verts, C.tris, C.numTris, p0, p1, p2, [&](const pdcol::V& nrm, const pdcol::V& hitp, float depth, int item) {
                    contactSet.insert(hitp, nrm, depth, 0, (unsigned)t * pdcol::ID_STRIDE + (unsigned)item);
                    ++contactCandidates;
                    // Car::onCollisionCallback (Car.cpp:960-1003)
                    flag = true;
                    const V3 n(nrm.x, nrm.y, nrm.z), hit(hitp.x, hitp.y, hitp.z);
                    const V3 posLocal = w2l(body, hit);
                    const V3 vel = pointVel(body, hit);
                    const float relSpeed = -((vel * n) * 3.6f);
                    const float fDamage = relSpeed * P->mechanicalDamageRate;
                    if (relSpeed > 0.0f && !noDamage) {
                        if (relSpeed * P->mechanicalDamageRate > 150.0f) blow = true;
                        const V3 vn = posLocal.get_norm();
                        int zone;
                        if (fabsf(vn.z) <= 0.70700002f) zone = (posLocal.x >= 0.0f) ? 2 : 3; else zone = (posLocal.z <= 0.0f) ? 1 : 0;
                        dmg[zone] = tmax(dmg[zone], fDamage);
                        dmg[4] = tmax(dmg[4], fDamage);
                    }
                });
        }
    }
    S.numContacts = contactSet.n;
    if (flag) S.collisionFlag = 1;
    if (blow) S.lifeLeft = -100.0f;   // Engine::blowUp (Engine.cpp:406-409)
    for (int i = 0; i < 5; ++i) {
        const float nv = tmax(S.damageZoneLevel[i], dmg[i]);
        if (fabsf(nv - S.damageZoneLevel[i]) > 0.001f) S.damageChanged = 1;
        S.damageZoneLevel[i] = nv;
    }
}

// ------------------------------------------------------------------------------------------------
void Car::carStep(float dt) {
    const pdb_car_params& Pm = *P;
    Body& body = w.bodies[PDB_BODY_CHASSIS];
    Body& tank = w.bodies[PDB_BODY_TANK];
    S.collisionFlag = 0; S.outOfTrackFlag = 0;
    if (physicsGUID == 0) {   // ERP/CFM switch (:426-451): `if (!physicsGUID)` -- the first car of a simulator only
        const V3 v = getVelocity(body);
        const float fVelSq = v.sqlen();
        const float erp = (fVelSq >= 1.0f) ? 0.3f : 0.9f;
        const float cfm = 0.0000001f;
        for (int j = 0; j < Pm.numJoints; ++j)
            if (Pm.joints[j].suspErp && w.joints[j].type == pdrb::JT_DBALL) { if (erp > 0.0f) w.joints[j].erp = erp; if (cfm > 0.0f) w.joints[j].cfm = cfm; }
    }
    {   // Car::updateAirPressure (Car.cpp:557-585; called from Car::step, :474): SlipStream::getSlipEffect of every OTHER car of the simulator (Sim/SlipStream.cpp:15-35)
        float fAirDensity = Pm.airDensity;
        const float slipStreamEffectGain = 1.0f;   // Car.h:135
        if (slipStreamEffectGain > 0.0f) {
            const V3 vPos = getPos(body);
            float fMinSlip = 1.0f;
            for (const pdb_slip_state& ss : otherSlips) {
                float fSlipE = 0;
                V3 vDelta = vPos - V3(ss.pos);
                const float fDeltaLen = vDelta.len();
                if (fDeltaLen < ss.length) {
                    vDelta.norm(fDeltaLen);
                    const float fDot = vDelta * V3(ss.dir);
                    if (fDot <= 0.7f) fSlipE = 0;
                    else fSlipE = (((1.0f - (fDeltaLen / ss.length)) * (fDot - 0.7f)) * 3.3333333f) * ss.effectGainMult;
                }
                const float fSlip = tclamp((1.0f - (fSlipE * slipStreamEffectGain)), 0.0f, 1.0f);
                if (fMinSlip > fSlip) fMinSlip = fSlip;
            }
            fAirDensity = ((fAirDensity - (fMinSlip * fAirDensity)) * (0.75f / slipStreamEffectGain)) + (fMinSlip * fAirDensity);
        }
        airDensityNow = fAirDensity;
    }
    controls.steer = tclamp(controls.steer, -1.0f, 1.0f);
    controls.clutch = tclamp(controls.clutch, 0.0f, 1.0f);
    controls.brake = tclamp(controls.brake, 0.0f, 1.0f);
    controls.handBrake = tclamp(controls.handBrake, 0.0f, 1.0f);
    controls.gas = tclamp(controls.gas, 0.0f, 1.0f);
    const float smoothSteerTarget = controls.steer;
    if (Pm.smoothSteer) {
        const float diff = smoothSteerTarget - S.smoothSteerValue;
        S.smoothSteerValue += diff * Pm.scoring.SmoothSteerSpeed * dt;
        controls.steer = S.smoothSteerValue;
    } else S.smoothSteerValue = smoothSteerTarget;
    {   // fuel (:476-489); fuelConsumptionRate = 0
        const float fRpmAbs = fabsf(engineRpm(*this));
        const double fNewFuel = S.fuel - (fRpmAbs * dt * S.gasUsage) * (0.0f + 1.0) * Pm.fuelConsumptionK * 0.001 * Pm.fuelConsumptionRate;
        S.fuel = fNewFuel;
        if (fNewFuel > 0.0f) S.fuelPressure = 1.0f; else { S.fuel = 0; S.fuelPressure = 0; }
    }
    float fSteerAngleSig = (Pm.steerLock * controls.steer) / Pm.steerRatio;
    if (!std::isfinite(fSteerAngleSig)) fSteerAngleSig = 0;
    finalSteerAngleSignal = fSteerAngleSig;
    bool bAllTyresLoaded = true;
    for (int i = 0; i < 4; ++i) if (S.tyre[i].load <= 0.0f) { bAllTyresLoaded = false; break; }
    autoClutchStep(*this, dt);
    {
        const float fSpeed = S.speed;
        const V3 av(body.avel);
        const float fAngVelSq = av.sqlen();
        if (fSpeed >= 0.5f || fAngVelSq >= 1.0f) S.sleepingFrames = 0;
        else {
            if (bAllTyresLoaded && (controls.gas <= 0.01f || controls.clutch <= 0.01f || S.currentGear == 1)) S.sleepingFrames++;
            else S.sleepingFrames = 0;
            if (S.sleepingFrames > 50) { body.stop(); tank.stop(); }
        }
    }
    {
        const V3 vBodyVel = getVelocity(body);
        const V3 vAccel = (vBodyVel - V3(S.lastVelocity)) * (1.0f / dt) * 0.10197838f;
        vBodyVel.store(S.lastVelocity);
        w2lN(body, vAccel).store(accG);
    }
    {   // stepThermalObjects (:624-634), ThermalObject::step (ThermalObject.cpp:11-23)
        const float fRpm = engineRpm(*this);
        float heat = 0;
        if (fRpm > (Pm.engMinimum * 0.8f)) {
            const float fLimiter = (float)limiterRpm(Pm);
            heat += ((((fRpm / fLimiter) * 20.0f) * controls.gas) + 85.0f);
        }
        const float fOneDivMass = 1.0f / Pm.waterTmass;
        const float fCool = 1.0f - (Pm.waterCoolSpeedK * S.speed);
        S.waterT += (((((fCool * Pm.ambientTemperature) - S.waterT) * fOneDivMass) * dt) * 0.2f);
        if (heat != 0.0f) S.waterT += ((((heat - S.waterT) * fOneDivMass) * dt) * 1.0f);
    }
    // ---- stepComponents ----
    {   // BrakeSystem::step (BrakeSystem.cpp:82-149)
        float fFrontBias = Pm.frontBias;
        if (Pm.ebbInternal) {   // EBBMode::Internal (:92-113): the front axle's share of the load, from the tyres' last loads
            const float fLoadFront = S.tyre[1].load + S.tyre[0].load;
            const float fLoadAWD = (S.tyre[3].load + S.tyre[2].load) + fLoadFront;
            bool bFlag = false;
            if (fLoadAWD != 0.0f) { if (S.speed * 3.6f > 10.0f) bFlag = true; }
            fFrontBias = bFlag ? tclamp(((fLoadFront / fLoadAWD) * Pm.ebbFrontMultiplier), 0.0f, 1.0f) : Pm.frontBias;
        } else if (Pm.ctrlEbb.count) fFrontBias = dynCtrlEval(*this, Pm.ctrlEbb);   // EBBMode::DynamicController (:90-93)
        fFrontBias = tclamp(fFrontBias, Pm.biasMin, Pm.biasMax);
        const float fBrakeInput = tmax(controls.brake, 0.0f);
        const float fBrakeTorq = (Pm.brakePower * Pm.brakePowerMultiplier) * fBrakeInput;
        ts[0].brakeTorque = fBrakeTorq * fFrontBias;
        ts[1].brakeTorque = fBrakeTorq * fFrontBias;
        float fRear = ((1.0f - fFrontBias) * fBrakeTorq) - 0.0f;
        if (fRear < 0.0f) fRear = 0;
        ts[2].brakeTorque = fRear; ts[3].brakeTorque = fRear;
        ts[2].handBrakeTorque = controls.handBrake * Pm.handBrakeTorque;
        ts[3].handBrakeTorque = controls.handBrake * Pm.handBrakeTorque;
        if (Pm.ctrlSteerBrake.count) {   // steerBrake (:136-143)
            const float fSteerBrake = dynCtrlEval(*this, Pm.ctrlSteerBrake);
            if (fSteerBrake >= 0.0f) ts[3].brakeTorque += fSteerBrake; else ts[2].brakeTorque -= fSteerBrake;
        }
        if (Pm.hasBrakeTemps) {   // BrakeSystem::stepTemps (:151-169)
            const float fAmbientTemp = Pm.ambientTemperature, fSpeed = kmh(S.speed);
            for (int i = 0; i < 4; ++i) {
                const pdb_brake_disc& d = Pm.discs[i];
                float t = S.brakeDiscT[i];
                ts[i].brakeTorque = curve(d.perfCurve, t) * ts[i].brakeTorque;
                const float fCool = ((fSpeed * d.coolSpeedFactor) + 1.0f) * d.coolTransfer;
                t += (((fAmbientTemp - t) * fCool) * dt);
                t += (((fabsf(S.tyre[i].angularVelocity) * (ts[i].brakeTorque * d.torqueK)) * 0.001f) * dt);
                S.brakeDiscT[i] = t;
            }
        }
    }
    for (int i = 0; i < 4; ++i) {
        if (Pm.susp[i].type == PDB_SUSP_STRUT) strutStep(Pm.susp[i], w, ts[i]);
        else if (Pm.susp[i].type == PDB_SUSP_DW) dwStep(Pm.susp[i], w, ts[i]);
        else if (Pm.susp[i].type == PDB_SUSP_ML) mlStep(Pm.susp[i], w, ts[i]);
        else axleStep(Pm.susp[i], w, ts[i]);
        S.suspTravel[i] = ts[i].travel;   // ISuspension status.travel: part of the record (controllers read it, the brake system's a tick later)
    }
    for (int i = 0; i < 4; ++i) tyreStep(*this, i, dt);
    for (int a = 0; a < 2; ++a) if (Pm.heave[a].k != 0.0f) heaveStep(Pm, Pm.heave[a], w, a * 2);   // Car.cpp:654-658
    if (Pm.numWings == 0) aeroDataStep(*this);   // AeroMap::step (AeroMap.cpp:85-97): the map's own drag and lift when there are no wings
    for (int wi = 0; wi < Pm.numWings; ++wi) wingStep(*this, wi);
    {   // SteeringSystem::step (SteeringSystem.cpp:17-24) -> setSteerLengthOffset (SuspensionStrut.cpp:340-350)
        const float steer = -finalSteerAngleSignal * Pm.steerLinearRatio;
        for (int j = 0; j < Pm.numJoints; ++j) {
            const int wi = Pm.joints[j].steerWheel;
            if (wi < 0) continue;
            const pdb_susp& su = Pm.susp[wi];
            const float sx = su.refPointSignX;
            const float d = 0.0f;
            const float offx = d + steer + (sx * su.toeOutLinear);
            const V3 carSteer(su.baseCarSteer[0] + offx, su.baseCarSteer[1], su.baseCarSteer[2]);
            // DistanceJointODE::reseatDistanceJointLocal (JointODE.cpp:77-89)
            float w1[3], w2[3];
            w.bodies[w.joints[j].b0].relPointPos(&carSteer.x, w1);
            w.bodies[w.joints[j].b1].relPointPos(su.tyreSteer, w2);
            w.dballSetAnchor1(j, w1);
            w.dballSetAnchor2(j, w2);
            w.joints[j].targetDistance = Pm.joints[j].distance;
        }
    }
    // AutoBlip::step (AutoBlip.cpp:51-75)
    if ((Pm.autoBlipActive || Pm.autoBlipElectronic) && (kmh(S.speed) > 5.0f)) {
        const double el = (S.physicsTime * 1000.0) - S.blipStartTime;
        if (el >= 0.0 && el < Pm.blipPerformTime && Pm.blipProfile.n == 4) {
            float fGas = controls.gas;
            const float prof = curve(Pm.blipProfile, (float)el);
            if (fGas <= prof) fGas = prof;
            float fNewGas = 1.0;
            if (fGas <= 1.0) { fNewGas = 0.0; if (fGas >= 0.0) fNewGas = fGas; }
            controls.gas = fNewGas;
        }
    }
    // AutoShifter::step (AutoShifter.cpp:31-117)
    if (Pm.autoShiftActive && S.currentGear) {
        if (!controls.gearUp && !controls.gearDn) {
            controls.gearDn = 0; controls.gearUp = 0;
            bool bIsSlipping = false;
            const float slip = (Pm.tractionType == 1) ? tmax(S.tyre[0].ndSlip, S.tyre[1].ndSlip) : tmax(S.tyre[2].ndSlip, S.tyre[3].ndSlip);
            if (slip > Pm.asSlipThreshold) { if (S.speed > 5.0f) bIsSlipping = true; }
            const bool changing = S.gearReqRequest != 0;
            if (!changing) {
                if ((controls.clutch > 0.99f || S.currentGear == 1) && !bIsSlipping) {
                    const int iEngineRpm = (int)engineRpm(*this);
                    if (iEngineRpm > Pm.asChangeUpRpm) {
                        if (S.currentGear < (Pm.numGears - 1) && controls.gas > 0.2f && S.asGasCutoff <= 0.0f) { controls.gearUp = true; S.asGasCutoff = Pm.asGasCutoffTime; }
                    }
                    const int iCurGear = S.currentGear;
                    int iChangeDnRpm;
                    if (iCurGear == 3) iChangeDnRpm = (int)(Pm.asChangeDnRpm * 0.65f); else iChangeDnRpm = Pm.asChangeDnRpm;
                    if (iEngineRpm < iChangeDnRpm && iCurGear > 2 && controls.clutch > 0.85f && S.asGasCutoff <= 0.0f) controls.gearDn = true;
                }
            }
            const bool bLowSpeed = S.speed < 2.0f;
            if (bLowSpeed && !(S.gearReqRequest != 0)) { if (controls.gas < 0.1f && S.asGasCutoff <= 0.0f && S.currentGear > 2) controls.gearDn = true; }
            const float fCutoff = S.asGasCutoff;
            if (fCutoff > 0.0f) { S.asGasCutoff = fCutoff - dt; controls.gas = 0.0f; }
        }
    }
    // GearChanger::step (GearChanger.cpp:18-40)
    if (controls.requestedGearIndex == -1) {
        if (controls.gearUp && !S.lastGearUp) gearUp(*this);
        if (controls.gearDn && !S.lastGearDn) gearDown(*this);
        S.lastGearUp = controls.gearUp ? 1 : 0;
        S.lastGearDn = controls.gearDn ? 1 : 0;
    } else {
        // Drivetrain::setCurrentGear(gearId, false) (Drivetrain.cpp:174-207); isGearboxLocked is never set on this path
        const int index = controls.requestedGearIndex;
        S.isGearGrinding = 0;
        if (index >= 0 && index < Pm.numGears && index != S.currentGear) {
            const double v8 = fabs(S.engineVel - Pm.gearRatio[index] * S.driveVel * Pm.finalRatio);
            const double v9 = ((controls.gas * locClutch * v8 - locClutch * v8) * Pm.controlsWindowGain + locClutch * v8) * (1.0 / (2.0 * 3.14159265358979323846)) * 60.0;
            if (index == 1 || v9 < S.validShiftRPMWindow) S.currentGear = index;
            else {
                S.isGearGrinding = 1;
                if (S.validShiftRPMWindow > 0.0) {
                    const double fRate = Pm.mechanicalDamageRate;
                    if (fRate > 0.0) S.validShiftRPMWindow -= Pm.damageRpmWindow * fRate * 0.003;
                }
            }
        }
    }
    drivetrainStep(*this, dt);
    // AntirollBar::step (AntirollBar.cpp:19-46)
    for (int a = 0; a < 2; ++a) {
        float k = Pm.arbK[a];
        if (Pm.ctrlArb[a].count) k = dynCtrlEval(*this, Pm.ctrlArb[a]);   // if (ctrl.ready) k = ctrl.eval() (AntirollBar.cpp:21-22)
        if (k > 0.0f) {
            const M44 mb = worldMatrix(body);
            const V3 vBodyM2(mb.m[4], mb.m[5], mb.m[6]);
            const M44 h0 = hubWorldMatrix(Pm, w, a * 2), h1 = hubWorldMatrix(Pm, w, a * 2 + 1);
            const V3 vHubWorld0(h0.m[12], h0.m[13], h0.m[14]), vHubWorld1(h1.m[12], h1.m[13], h1.m[14]);
            const V3 vHubLoc0 = w2l(body, vHubWorld0), vHubLoc1 = w2l(body, vHubWorld1);
            const float fDelta = vHubLoc1.y - vHubLoc0.y;
            const float fDeltaK = fDelta * k;
            const V3 vForce = vBodyM2.get_norm() * fDeltaK;
            hubAddForceAtPos(w, Pm.susp[a * 2], vForce, vHubWorld0);
            hubAddForceAtPos(w, Pm.susp[a * 2 + 1], vForce * -1.0f, vHubWorld1);
            const V3 f0(0, -fDeltaK, 0), f1(0, fDeltaK, 0);
            body.addRelForceAtRelPos(&f0.x, &vHubLoc0.x);
            body.addRelForceAtRelPos(&f1.x, &vHubLoc1.x);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Track queries (Sim/Track.cpp:436-699), Car::postStep (Car.cpp:685-865), ScoringSystem (ScoringSystem.cpp:114-398)
// ------------------------------------------------------------------------------------------------
static bool lineIntersect(float p0x, float p0y, float p1x, float p1y, float p2x, float p2y, float p3x, float p3y, float& ix, float& iy) {
    const float s1x = p1x - p0x, s1y = p1y - p0y, s2x = p3x - p2x, s2y = p3y - p2y;
    const float s = (-s1y * (p0x - p2x) + s1x * (p0y - p2y)) / (-s2x * s1y + s1x * s2y);
    const float t = (s2x * (p0y - p2y) - s2y * (p0x - p2x)) / (-s2x * s1y + s1x * s2y);
    if (s >= 0 && s <= 1 && t >= 0 && t <= 1) { ix = p0x + (t * s1x); iy = p0y + (t * s1y); return true; }
    return false;
}

static float rayCastTrackBounds(Car& c, const V3& pos, const V3& dir, float maxDistance) {
    const TrackData& T = *c.T;
    if (maxDistance <= 0.0f) maxDistance = T.h->hashCellSize;
    float result = maxDistance;
    if ((V3(c.S.pointCachePos) - pos).sqlen() > 1.0f * 1.0f) {
        pos.store(c.S.pointCachePos);
        // VertexHash::queryNeighbours (Core/VertexHash.h:45-88): every fat point with |p - origin|^2 < maxDistance^2
        // (27-cell neighbourhood of a 50 m grid covers the 50 m radius); canonical order = ascending id
        c.nearby.clear();
        const float md2 = maxDistance * maxDistance;
        for (int id = 0; id < T.h->numFat; ++id) { const V3 p(T.fat + 15 * id); if ((pos - p).sqlen() < md2) c.nearby.push_back(id); }
    }
    if (!c.nearby.empty()) {
        const V3 rayEnd = pos + dir * (maxDistance * 1.1f);
        const float ax = pos.x, ay = pos.z, bx = rayEnd.x, by = rayEnd.z;
        float ix = 0, iy = 0, bestDist = FLT_MAX;
        bool interFlag = false;
        const int maxPoints = T.h->numFat;
        for (int id : c.nearby) {
            const int other = id + 1 < maxPoints ? id + 1 : 0;
            for (int side = 0; side < 2; ++side) {
                const float* A = T.fat + 15 * id + (side ? 6 : 3);
                const float* B = T.fat + 15 * other + (side ? 6 : 3);
                if (lineIntersect(ax, ay, bx, by, A[0], A[2], B[0], B[2], ix, iy)) {
                    const float dx = ax - ix, dy = ay - iy;
                    bestDist = tmin(bestDist, sqrtf(dx * dx + dy * dy));
                    interFlag = true;
                }
            }
        }
        if (interFlag) result = bestDist;
    }
    return result;
}

static int pointIdAtDistance(const TrackData& T, float distanceNorm) {
    const int n = T.h->numFat;
    if (!n) return 0;
    if (distanceNorm < 0.0f) distanceNorm += 1.0f; else if (distanceNorm > 1.0f) distanceNorm -= 1.0f;
    return (int)(size_t)(tclamp(distanceNorm, 0.0f, 1.0f) * (float)(n - 1));
}
static V3 trackDirAtDistance(const TrackData& T, float distanceNorm) {
    const int id = pointIdAtDistance(T, distanceNorm);
    if (id < T.h->numFat) return V3(T.fat + 15 * id + 12);
    return V3(0, 0, 0);
}

void Car::postStep(float dt) {
    const pdb_car_params& Pm = *P;
    const TrackData& Tk = *T;
    Body& body = w.bodies[PDB_BODY_CHASSIS];
    {   // slipStream->setPosition(body position, body velocity) (Car.cpp:692-694, Sim/SlipStream.cpp:37-47; the triangle's other two corners are never read)
        const V3 vel = getVelocity(body), pos = getPos(body);
        const float fVelLen = vel.len();
        const V3 dir = vel.get_norm(fVelLen) * -1.0f;
        slip.pos[0] = pos.x; slip.pos[1] = pos.y; slip.pos[2] = pos.z;
        slip.dir[0] = dir.x; slip.dir[1] = dir.y; slip.dir[2] = dir.z;
        slip.length = (fVelLen * 0.25f) * Pm.slipSpeedFactorMult;
        slip.effectGainMult = Pm.slipEffectGainMult;
    }
    // updateTrackLocator (Car.cpp:717-771)
    for (int r = 0; r < PDB_NUM_PROBES; ++r) {
        const V3 rp(0, 0, 0), rd(Pm.probeDir[r]);
        const float rl = Pm.probeLen[r];
        const V3 rayStart = l2w(body, rp);
        const V3 rayEnd = l2w(body, rp + rd * rl);
        probeHits[r] = rayCastTrackBounds(*this, rayStart, (rayEnd - rayStart).get_norm(), rl);
    }
    const V3 bodyPos = getPos(body);
    int bestPoint = 0;
    {
        float bestDistSq = FLT_MAX;
        for (int id : nearby) { const V3 p(Tk.fat + 15 * id); const float d = (p - bodyPos).sqlen(); if (bestDistSq > d) { bestDistSq = d; bestPoint = id; } }
    }
    if (S.nearestTrackPointId != bestPoint) { S.oldTrackPointId = S.nearestTrackPointId; S.nearestTrackPointId = bestPoint; S.lastTrackPointTimestamp = (float)S.physicsTime; }
    S.oldTrackLocation = S.trackLocation;
    S.trackLocation = 0;
    const int numPoints = Tk.h->numFat;
    if (bestPoint >= 0 && bestPoint < numPoints) {
        // Track::getDistanceAlongSplineAtLocation (Track.cpp:582-699) -> Spline3d::find_nearest_point (Spline3d.cpp:34-75)
        if (numPoints >= 5) {
            int prevId = bestPoint - 1; if (prevId < 0) prevId = numPoints - 1;
            int nextId = bestPoint + 1; if (nextId >= numPoints) nextId = 0;
            int prevId2 = prevId - 1; if (prevId2 < 0) prevId2 = numPoints - 1;
            int nextId2 = nextId + 1; if (nextId2 >= numPoints) nextId2 = 0;
            int seg1 = prevId2 * Tk.h->interpolateStep, seg2 = nextId2 * Tk.h->interpolateStep;
            const int npoints = Tk.h->numNodes;
            if (seg1 >= npoints) seg1 = 0;
            if (seg2 >= npoints) seg2 = 0;
            if (!seg1 && !seg2) seg2 = npoints;
            int best_id = 0; float best_dist = FLT_MAX, spline_dist = 0; bool found = false;
            int id = seg1;
            while (id != seg2) {
                const V3 pt(Tk.nodes + 3 * id);
                const float d = (bodyPos - pt).sqlen();
                if (best_dist >= d) { best_dist = d; best_id = id; spline_dist = Tk.nodeDist[id]; found = true; }
                ++id;
                if (id >= npoints) id = 0;
            }
            int infoId = best_id; float infoDist = spline_dist;
            if (!found) {
                // fallback segment trace (Track.cpp:607-676)
                const int ids[5] = {prevId2, prevId, bestPoint, nextId, nextId2};
                int bestPointId = 0; float bestDist = FLT_MAX, splineDist = 0;
                for (int i = 0; i + 1 < 5; ++i) {
                    const V3 s1(Tk.fat + 15 * ids[i]), s2(Tk.fat + 15 * ids[i + 1]);
                    const float slen = (s2 - s1).len();
                    const V3 n = (s2 - s1) / slen;
                    for (float tracePos = 0; tracePos <= slen; tracePos += 0.02f) {
                        const V3 p = s1 + n * tracePos;
                        const float d = (bodyPos - p).sqlen();
                        if (bestDist >= d) { bestDist = d; bestPointId = ids[i]; splineDist = Tk.fatDist[bestPointId] + tracePos; }
                    }
                }
                infoId = bestPointId; infoDist = splineDist;
            }
            S.splinePointId = infoId;
            S.trackLocation = tclamp(infoDist / Tk.h->computedTrackLength, 0.0f, 1.0f);
        }
        const M44 bm = worldMatrix(body);
        // (vec3f(0,0,1) * bodyR).get_norm()  (Core/Math.h:196-203 with M41..43 = 0)
        const V3 f(0, 0, 1);
        const V3 bodyFrontDir = V3(0.0f + (f * V3(bm.m[0], bm.m[4], bm.m[8])), 0.0f + (f * V3(bm.m[1], bm.m[5], bm.m[9])), 0.0f + (f * V3(bm.m[2], bm.m[6], bm.m[10]))).get_norm();
        const V3 bodyVelDir = getVelocity(body).get_norm();
        const V3 fwd(Tk.fat + 15 * bestPoint + 12);
        S.bodyVsTrack = bodyFrontDir * fwd;
        if (kmh(S.speed) > 3.0f) S.velocityVsTrack = bodyVelDir * fwd; else S.velocityVsTrack = 0.0f;
    }
    // updateLookAhead (Car.cpp:775-798)
    {
        const V3 up(0, 1, 0);
        const V3 curTrackDir = trackDirAtDistance(Tk, S.trackLocation);
        const M44 bm = worldMatrix(body);
        const V3 f(0, 0, 1);
        const V3 bodyFrontDir = V3(0.0f + (f * V3(bm.m[0], bm.m[4], bm.m[8])), 0.0f + (f * V3(bm.m[1], bm.m[5], bm.m[9])), 0.0f + (f * V3(bm.m[2], bm.m[6], bm.m[10]))).get_norm();
        const float driveDir = signf_(bodyFrontDir * curTrackDir);
        for (int i = 0; i < PDB_NUM_LOOKAHEAD; ++i) {
            const float distanceNorm = S.trackLocation + ((Pm.lookAheadStep * (float)(i + 1)) / Tk.h->computedTrackLength) * driveDir;
            const V3 dir = trackDirAtDistance(Tk, distanceNorm);
            lookAhead[i] = m_atan2f(dir.cross(curTrackDir) * up, curTrackDir * dir);
        }
    }
    // ScoringSystem::step: computeDriftScore then computeAgentReward
    {
        // validateDrift (ScoringSystem.cpp:338-371)
        bool bInvalid = true;
        int nDirty = 0;
        for (int i = 0; i < 4; ++i) nDirty += (ts[i].surface >= 0 && Tk.surfaces[ts[i].surface].dirtAdditiveK > 0.001f) ? 1 : 0;
        if (nDirty <= 2) { if (kmh(S.speed) >= 20.0f) { if (!S.damageChanged && S.currentGear) bInvalid = false; } }
        if (bInvalid) S.driftInvalid = 1;
        // getBetaRad (Car.cpp:1472-1484)
        const V3 lvel = w2lN(body, getVelocity(body));
        float fBeta;
        {
            V3 vel = lvel;
            const float fLen = vel.len();
            if (fLen != 0.0f) vel.x /= fLen;
            if (vel.x <= -1.0f || vel.x >= 1.0f) fBeta = 1.5707964f; else fBeta = m_asinf(vel.x);
            fBeta = fabsf(fBeta);
        }
        const float fSpeedKmh = kmh(S.speed);
        bool done = false;
        if (fSpeedKmh > 20.0f && fBeta > 0.13089749f) {
            const V3 v = lvel;
            if (!S.drifting) { S.lastDriftDirection = signf_(v.x); S.driftComboCounter = 1; S.driftInvalid = 0; S.instantDrift = 0.0f; }
            S.currentDriftAngle = fBeta - 0.13089749f;
            float fSpeedMult = (fSpeedKmh - 20.0f) * 0.015384615f;
            fSpeedMult = tclamp(fSpeedMult, 0.0f, 2.0f);
            S.currentSpeedMultiplier = fSpeedMult;
            int nDrifty = 0;
            for (int i = 0; i < 4; ++i) {
                const bool e = ts[i].surface >= 0 && fabsf(S.tyre[i].angularVelocity) > 4.0 && fabsf(S.tyre[i].slipRatio) > 0.8f && S.tyre[i].load > 10.0 &&
                               Tk.surfaces[ts[i].surface].gripMod >= 0.9f;
                nDrifty += e ? 1 : 0;
            }
            S.driftExtreme = nDrifty > 1;
            float fDelta = fSpeedMult * S.currentDriftAngle;
            if (S.driftExtreme) fDelta *= 2.0f;
            S.instantDriftDelta = fDelta;
            S.instantDrift += fDelta;
            if (fabsf(v.x) > 4.0f) {
                const float fDir = signf_(v.x);
                if (S.lastDriftDirection != fDir && fBeta > 0.26179498f) { S.instantDrift += 50.0f; S.driftComboCounter++; S.lastDriftDirection = fDir; }
            }
            S.drifting = 1;
            S.driftStraightTimer = 0.0f;
        }
        auto resetDrift = [&]() { S.currentDriftAngle = 0.0; S.currentSpeedMultiplier = 0.0; S.driftExtreme = 0; S.drifting = 0; S.instantDrift = 0.0; S.driftComboCounter = 0; };
        if (S.drifting) {
            if (fSpeedKmh > 20.0f && fBeta < 0.065448746f) S.driftStraightTimer += dt; else S.driftStraightTimer = 0.0f;
            if (S.driftInvalid) { resetDrift(); done = true; }
            else if (S.driftStraightTimer > 1.0f) { S.driftComboCounter = 0; S.driftPoints += S.instantDrift; S.drifting = 0; S.instantDrift = 0.0f; }
        }
        if (!done && S.driftInvalid) resetDrift();
    }
    {   // computeAgentReward (ScoringSystem.cpp:129-248)
        const pdb_scoring& sv = Pm.scoring;
        float reward = 0.0f;
        const float curRpm = carEngineRpm(*this);
        const float maxRpm = (float)limiterRpm(Pm);
        if (S.oldPointId < S.nearestTrackPointId || (S.nearestTrackPointId == 0 && S.oldPointId != S.nearestTrackPointId)) { S.oldPointId = S.nearestTrackPointId; reward += sv.TravelBonus; }
        if (S.oldSplinePointId < S.splinePointId || (S.splinePointId == 0 && S.oldSplinePointId != S.splinePointId)) { S.oldSplinePointId = S.splinePointId; reward += sv.TravelSplineBonus; }
        reward += sv.DriftBonus * S.instantDriftDelta;
        reward += sv.SpeedBonus * linscalef(kmh(S.speed), sv.MinBonusSpeed, sv.MaxBonusSpeed, 0.0f, 1.0f);
        reward += sv.ThrottleBonus * linscalef(controls.gas, 0.0f, 1.0f, 0.0f, 1.0f);
        reward += sv.EngineRpmBonus * linscalef(curRpm, 0.0f, maxRpm, 0.0f, 1.0f);
        if (carEngineRpm(*this) < sv.StallRpm) reward -= sv.StallPenalty;
        if (S.isGearGrinding) reward -= sv.GearGrindPenalty;
        {
            float closestProbe = FLT_MAX;
            for (int i = 0; i < PDB_NUM_PROBES; ++i) { const float dist = probeHits[i]; if (closestProbe > dist && dist > 0.0f) closestProbe = dist; }
            if (closestProbe < sv.ApproachDistance) reward -= sv.ObstApproachPenalty * (1.0f - linscalef(closestProbe, sv.CriticalDistance, sv.ApproachDistance, 0.0f, 1.0f));
        }
        // setCarAutoTeleport (PyProjectD.cpp:292-295): Car::teleportByMode inside the reward computation (ScoringSystem.cpp:194-225).
        // The teleport itself is the PRODUCT's host function, handed in by the test (autoTeleportHook), so that this path pins it.
        auto autoTeleport = [&]() {
            if (!autoTeleportHook) return;
            storeState();
            autoTeleportHook(&S, (P->autoTeleport >> 2) & 3);
            const pdb_dyn_state st = S;
            loadState(st);
            for (int i = 0; i < 4; ++i) ts[i].feedbackTorque = 0;   // Tyre::reset (Tyre.cpp:419): per-tick scratch here, seen by the probe only
        };
        if (S.collisionFlag) { reward -= sv.CollisionPenalty; if (P->autoTeleport & 1) autoTeleport(); }
        const int tp = S.nearestTrackPointId;
        if (tp >= 0 && tp < Tk.h->numFat) {
            const V3 center(Tk.fat + 15 * tp + 9);
            if ((getPos(body) - center).len() > Tk.h->computedTrackWidth * sv.OutOfTrackThreshold) { S.outOfTrackFlag = 1; reward -= sv.OffTrackPenalty; if (P->autoTeleport & 2) autoTeleport(); }
            const float x = S.bodyVsTrack;
            const float thresh = tclamp(sv.DirectionThreshold, 0.1f, 1.0f);
            if (x > thresh) reward += sv.DirectionBonus * linscalef(x, thresh, 1.0f, 0.0f, 1.0f);
            else reward -= sv.DirectionPenalty * (1.0f - linscalef(x, -1.0f, thresh, 0.0f, 1.0f));
        }
        S.stepReward = reward;
        S.totalReward += reward;
    }
    S.oldCollisionFlag = S.collisionFlag;
    for (int i = 0; i < 4; ++i) { const M44 hm = hubWorldMatrix(Pm, w, i); memcpy(ts[i].hubMatrix, hm.m, sizeof(hm.m)); }
}

// PyProjectD.cpp:297-305 (setCarControls) + :160-180 (stepSimulator) + Simulator::step (Simulator.cpp:168-201)
void Car::step(float steer, float gas, float dt, double dtD) {
    pdb_controls c;
    memset(&c, 0, sizeof(c));
    c.steer = steer; c.gas = gas; c.isShifterSupported = 1; c.requestedGearIndex = -1;
    stepControls(c, dt, dtD);
}
void Car::stepControls(const pdb_controls& c, float dt, double dtD) {
    controls = c;
    stepTime = S.physicsTime;
    S.speed = getVelocity(w.bodies[PDB_BODY_CHASSIS]).len();   // Car::stepPreCacheValues (Car.cpp:414-417)
    carStep(dt);
    collisionStep();   // PhysicsEngineODE::step: collisionStep, then dWorldStep (PhysicsEngineODE.cpp:216-224)
    w.contacts.resize(S.numContacts);
    for (int i = 0; i < S.numContacts; ++i) memcpy(&w.contacts[i], &contactSet.c[i], sizeof(pdrb::ContactJoint));
    w.step(dt);
    postStep(dt);
    storeState();
    S.physicsTime += dtD;
}

// Car::updateCarState (Car.cpp:802-865); timestamp is the pre-increment physicsTime
void Car::fillCarState(pdb_car_state& cs) const {
    memset(&cs, 0, sizeof(cs));
    const Body& body = w.bodies[PDB_BODY_CHASSIS];
    cs.carId = 0; cs.simId = 0;
    cs.timestamp = (float)stepTime;
    cs.controls = controls;
    cs.collisionFlag = S.collisionFlag; cs.outOfTrackFlag = S.outOfTrackFlag; cs.trackPointId = S.nearestTrackPointId;
    cs.lastTrackPointTimestamp = S.lastTrackPointTimestamp; cs.trackLocation = S.trackLocation;
    cs.bodyVsTrack = S.bodyVsTrack; cs.velocityVsTrack = S.velocityVsTrack;
    cs.engineRPM = carEngineRpm(*this); cs.speedMS = S.speed; cs.gear = S.currentGear; cs.gearGrinding = S.isGearGrinding ? 1 : 0;
    const M44 bm = worldMatrix(body);
    memcpy(cs.bodyMatrix, bm.m, sizeof(bm.m));
    cs.bodyPos[0] = bm.m[12]; cs.bodyPos[1] = bm.m[13]; cs.bodyPos[2] = bm.m[14];
    {   // mat44f::getEulerAngles (Core/Math.cpp:60-85)
        float rx = m_atan2f(-bm.m[8], bm.m[10]);
        float v7 = 1, v8 = bm.m[9];
        if (v8 > 1.0 || (v7 = -1, v8 < -1.0)) v8 = v7;
        const float ry = m_asinf(v8);
        float v10, v11;
        if (bm.m[1] == 0.0f && bm.m[5] == 0.0f) { v11 = bm.m[4]; v10 = bm.m[0]; rx = 0.0f; } else { v11 = -bm.m[1]; v10 = bm.m[5]; }
        const float rz = m_atan2f(v11, v10);
        cs.bodyEuler[0] = ry * -57.295779513082323f; cs.bodyEuler[1] = rx * -57.295779513082323f; cs.bodyEuler[2] = rz * -57.295779513082323f;
    }
    memcpy(cs.accG, accG, 12);
    const V3 v = getVelocity(body);
    v.store(cs.velocity);
    w2lN(body, v).store(cs.localVelocity);
    memcpy(cs.angularVelocity, body.avel, 12);
    w2lN(body, V3(body.avel)).store(cs.localAngularVelocity);
    for (int i = 0; i < 4; ++i) {
        memcpy(cs.hubMatrix[i], ts[i].hubMatrix, 64);
        memcpy(cs.tyreContacts[i], S.tyre[i].contactPoint, 12);
        cs.tyreLoad[i] = S.tyre[i].load; cs.tyreAngularSpeed[i] = S.tyre[i].angularVelocity;
        cs.tyreSlipRatio[i] = S.tyre[i].slipRatio; cs.tyreNdSlip[i] = S.tyre[i].ndSlip;
    }
    for (int i = 0; i < PDB_NUM_PROBES; ++i) cs.probes[i] = probeHits[i];
    for (int i = 0; i < PDB_NUM_LOOKAHEAD; ++i) cs.lookAhead[i] = lookAhead[i];
    cs.stepReward = S.stepReward; cs.totalReward = S.totalReward;
}

// pyprojectd/projectd_env.py:237-275 observation order; :182-199 flags
void Car::fillStepOut(pdb_step_out& o) const {
    pdb_car_state cs;
    fillCarState(cs);
    int k = 0;
    for (int i = 0; i < 3; ++i) o.obs[k++] = cs.localVelocity[i];
    for (int i = 0; i < 3; ++i) o.obs[k++] = cs.localAngularVelocity[i];
    for (int i = 0; i < 4; ++i) o.obs[k++] = cs.tyreNdSlip[i];
    o.obs[k++] = cs.bodyVsTrack; o.obs[k++] = cs.velocityVsTrack;
    for (int i = 0; i < 5; ++i) o.obs[k++] = cs.lookAhead[i];
    for (int i = 0; i < 7; ++i) o.obs[k++] = cs.probes[i];
    o.reward = cs.stepReward;
    o.flags = (cs.collisionFlag ? 1 : 0) | (cs.outOfTrackFlag ? 2 : 0) | (((double)cs.lastTrackPointTimestamp + 5.0 < (double)cs.timestamp) ? 4 : 0);
    {   // fault bit (include/pdb_types.h): non-finite chassis pose / velocity
        const Body& b = w.bodies[PDB_BODY_CHASSIS];
        const float chk = ((b.pos[0] + b.pos[1]) + (b.pos[2] + b.q[0])) + ((b.q[1] + b.q[2]) + (b.q[3] + b.lvel[0])) + ((b.lvel[1] + b.lvel[2]) + (b.avel[0] + b.avel[1])) + b.avel[2];
        if (!std::isfinite(chk)) o.flags |= 32;
    }
}

void Car::fillProbe(pdoracle::Probe& Pr) const {
    const double tPre = S.physicsTime;   // the harness samples after `physicsTime += dt`
    Pr.p("time", tPre);
    const bool legacy = P->susp[0].type == PDB_SUSP_STRUT && P->susp[2].type == PDB_SUSP_AXLE;
    const char* bnl[7] = {"chassis", "tank", "axle", "hub0", "strut0", "hub1", "strut1"};
    char nm[96], bname[16];
    for (int i = 0; i < P->numBodies; ++i) {
        const Body& b = w.bodies[i];
        if (!legacy) snprintf(bname, sizeof(bname), "body%d", i);
        const char* bn = legacy ? bnl[i] : bname;
        snprintf(nm, sizeof(nm), "%s.pos", bn); Pr.p3(nm, b.pos);
        snprintf(nm, sizeof(nm), "%s.q", bn); Pr.pn(nm, b.q, 4);
        snprintf(nm, sizeof(nm), "%s.R", bn); Pr.pn(nm, b.R, 9);
        snprintf(nm, sizeof(nm), "%s.lvel", bn); Pr.p3(nm, b.lvel);
        snprintf(nm, sizeof(nm), "%s.avel", bn); Pr.p3(nm, b.avel);
    }
    Pr.p("ctrl.steer", controls.steer); Pr.p("ctrl.clutch", controls.clutch); Pr.p("ctrl.brake", controls.brake);
    Pr.p("ctrl.handBrake", controls.handBrake); Pr.p("ctrl.gas", controls.gas); Pr.p("ctrl.gearUp", controls.gearUp); Pr.p("ctrl.gearDn", controls.gearDn);
    Pr.p("car.finalSteerAngleSignal", finalSteerAngleSignal);
    Pr.p("car.smoothSteerValue", S.smoothSteerValue);
    Pr.p3("car.accG", accG);
    Pr.p("car.sleepingFrames", S.sleepingFrames);
    Pr.p("car.speed", S.speed);
    Pr.p3("car.lastVelocity", S.lastVelocity);
    Pr.p("car.waterT", S.waterT);
    Pr.p("car.fuel", S.fuel);
    Pr.p("aero.airDensity", airDensityNow != 0.0f ? airDensityNow : P->airDensity);
    for (int i = 0; i < 4; ++i) {
        const pdb_tyre_state& st = S.tyre[i];
        const TyreScratch& sc = ts[i];
#define TP(field, val) snprintf(nm, sizeof(nm), "tyre%d." field, i); Pr.p(nm, val)
        TP("angularVelocity", st.angularVelocity); TP("slipAngleRAD", st.slipAngleRAD); TP("slipRatio", st.slipRatio);
        TP("ndSlip", st.ndSlip); TP("load", st.load); TP("Fx", st.Fx); TP("Fy", st.Fy); TP("Mz", st.Mz);
        TP("isLocked", st.isLocked ? 1 : 0); TP("dirtyLevel", st.dirtyLevel); TP("flatSpot", st.flatSpot);
        TP("inflation", st.inflation); TP("pressureDynamic", st.pressureDynamic); TP("pressureStatic", P->tyre[i].pressureStatic);
        TP("loadedRadius", st.loadedRadius); TP("effectiveRadius", st.effectiveRadius); TP("liveRadius", sc.liveRadius);
        TP("camberRAD", st.camberRAD); TP("D", st.D); TP("Dx", sc.Dx); TP("Dy", sc.Dy); TP("depth", sc.depth);
        TP("distToGround", sc.distToGround); TP("feedbackTorque", sc.feedbackTorque);
        TP("rollingResistence", sc.rollingResistence); TP("thermalInput", sc.thermalInput);
        TP("slipFactor", sc.slipFactor); TP("virtualKM", st.virtualKM); TP("wearMult", sc.wearMult);
        TP("grain", 0.0); TP("blister", 0.0);
        TP("localMX", st.localMX); TP("oldAngularVelocity", st.oldAngularVelocity);
        TP("totalHubVelocity", sc.totalHubVelocity); TP("slidingVelocityX", sc.slidingVelocityX);
        TP("slidingVelocityY", sc.slidingVelocityY); TP("roadVelocityX", sc.roadVelocityX);
        TP("brakeTorque", sc.brakeTorque); TP("handBrakeTorque", sc.handBrakeTorque);
        snprintf(nm, sizeof(nm), "tyre%d.contactPoint", i); Pr.p3(nm, st.contactPoint);
        snprintf(nm, sizeof(nm), "tyre%d.unmodifiedContactPoint", i); Pr.p3(nm, st.unmodifiedContactPoint);
        snprintf(nm, sizeof(nm), "tyre%d.contactNormal", i); Pr.p3(nm, st.contactNormal);
        TP("coreTemp", st.coreTemp); TP("phase", st.phase); TP("thermalMultD", st.thermalMultD);
        TP("practicalTemp", st.practicalTemp);
        for (int k = 0; k < 36; ++k) { snprintf(nm, sizeof(nm), "tyre%d.T[%d]", i, k); Pr.p(nm, st.T[k]); }
        TP("susp.travel", sc.travel); TP("susp.damperSpeedMS", sc.damperSpeedMS);
#undef TP
    }
    Pr.p("dt.engine.velocity", S.engineVel); Pr.p("dt.drive.velocity", S.driveVel);
    Pr.p("dt.outShaftL.velocity", S.outShaftLVel); Pr.p("dt.outShaftR.velocity", S.outShaftRVel);
    Pr.p("dt.rootVelocity", S.rootVelocity); Pr.p("dt.locClutch", locClutch);
    Pr.p("dt.currentClutchTorque", currentClutchTorque); Pr.p("dt.ratio", ratio); Pr.p("dt.lastRatio", S.lastRatio);
    Pr.p("dt.cutOff", S.cutOff); Pr.p("dt.totalTorque", totalTorque); Pr.p("dt.currentGear", S.currentGear);
    Pr.p("dt.isGearGrinding", S.isGearGrinding ? 1 : 0); Pr.p("dt.clutchOpenState", S.clutchOpenState ? 1 : 0);
    Pr.p("dt.gearRequest.request", S.gearReqRequest); Pr.p("dt.gearRequest.timeAccumulator", S.gearReqTimeAccumulator);
    Pr.p("dt.gearRequest.timeout", S.gearReqTimeout); Pr.p("dt.gearRequest.requestedGear", S.gearReqRequestedGear);
    Pr.p("dt.validShiftRPMWindow", S.validShiftRPMWindow);
    Pr.p("eng.outTorque", engOutTorque); Pr.p("eng.limiterOn", S.limiterOn); Pr.p("eng.lifeLeft", S.lifeLeft);
    Pr.p("eng.gasUsage", gasUsage); Pr.p("eng.fuelPressure", S.fuelPressure); Pr.p("eng.turboBoost", turboBoost);
    Pr.p("ac.clutchValueSignal", S.acClutchValueSignal);
    Pr.p("ac.seq.currentTime", S.acSeqCurrentTime);
    Pr.p("ac.seq.isDone", S.acSeqIsDone ? 1 : 0);
    Pr.p("ac.seq.count", S.acSeqActive ? 4 : 0);
    Pr.p("ab.blipStartTime", S.blipStartTime);
    Pr.p("as.gasCutoff", S.asGasCutoff);
    Pr.p("gc.lastGearUp", S.lastGearUp ? 1 : 0); Pr.p("gc.lastGearDn", S.lastGearDn ? 1 : 0);
    for (int i = 0; i < P->numWings; ++i) {
#define WP(field, val) snprintf(nm, sizeof(nm), "wing%d." field, i); Pr.p(nm, val)
        WP("aoa", ws[i].aoa); WP("yawAngle", ws[i].yawAngle); WP("cd", ws[i].cd); WP("cl", ws[i].cl);
        WP("dragKG", ws[i].dragKG); WP("liftKG", ws[i].liftKG); WP("groundHeight", ws[i].groundHeight);
#undef WP
    }
    Pr.p("trk.nearestTrackPointId", S.nearestTrackPointId); Pr.p("trk.oldTrackPointId", S.oldTrackPointId);
    Pr.p("trk.splinePointId", S.splinePointId); Pr.p("trk.lastTrackPointTimestamp", S.lastTrackPointTimestamp);
    Pr.p("trk.trackLocation", S.trackLocation); Pr.p("trk.oldTrackLocation", S.oldTrackLocation);
    Pr.p("trk.bodyVsTrack", S.bodyVsTrack); Pr.p("trk.velocityVsTrack", S.velocityVsTrack);
    for (int i = 0; i < 7; ++i) { snprintf(nm, sizeof(nm), "trk.probe[%d]", i); Pr.p(nm, probeHits[i]); }
    for (int i = 0; i < 5; ++i) { snprintf(nm, sizeof(nm), "trk.lookAhead[%d]", i); Pr.p(nm, lookAhead[i]); }
    Pr.p("sc.stepReward", S.stepReward); Pr.p("sc.totalReward", S.totalReward);
    Pr.p("sc.oldPointId", S.oldPointId); Pr.p("sc.oldSplinePointId", S.oldSplinePointId);
    Pr.p("sc.drifting", S.drifting ? 1 : 0); Pr.p("sc.driftExtreme", S.driftExtreme ? 1 : 0);
    Pr.p("sc.driftInvalid", S.driftInvalid ? 1 : 0); Pr.p("sc.currentDriftAngle", S.currentDriftAngle);
    Pr.p("sc.currentSpeedMultiplier", S.currentSpeedMultiplier); Pr.p("sc.lastDriftDirection", S.lastDriftDirection);
    Pr.p("sc.driftStraightTimer", S.driftStraightTimer); Pr.p("sc.instantDriftDelta", S.instantDriftDelta);
    Pr.p("sc.instantDrift", S.instantDrift); Pr.p("sc.driftPoints", S.driftPoints);
    Pr.p("sc.driftComboCounter", S.driftComboCounter);
    Pr.p("car.collisionFlag", S.collisionFlag ? 1 : 0); Pr.p("car.outOfTrackFlag", S.outOfTrackFlag ? 1 : 0);
    for (int i = 0; i < 5; ++i) { char nm[40]; snprintf(nm, sizeof(nm), "car.damageZoneLevel%d", i); Pr.p(nm, S.damageZoneLevel[i]); }
    pdb_car_state cs;
    fillCarState(cs);
    Pr.p("cs.timestamp", cs.timestamp); Pr.p("cs.engineRPM", cs.engineRPM); Pr.p("cs.speedMS", cs.speedMS);
    Pr.p("cs.gear", cs.gear); Pr.p("cs.gearGrinding", cs.gearGrinding);
    Pr.p("cs.trackPointId", cs.trackPointId); Pr.p("cs.lastTrackPointTimestamp", cs.lastTrackPointTimestamp);
    Pr.p3("cs.bodyEuler", cs.bodyEuler); Pr.p3("cs.accG", cs.accG); Pr.p3("cs.velocity", cs.velocity);
    Pr.p3("cs.localVelocity", cs.localVelocity); Pr.p3("cs.angularVelocity", cs.angularVelocity);
    Pr.p3("cs.localAngularVelocity", cs.localAngularVelocity);
    for (int i = 0; i < 4; ++i) { snprintf(nm, sizeof(nm), "cs.hubMatrix%d", i); Pr.pn(nm, cs.hubMatrix[i], 16); }
    for (int i = 0; i < 4; ++i) { snprintf(nm, sizeof(nm), "cs.tyreContacts%d", i); Pr.p3(nm, cs.tyreContacts[i]); }
    Pr.pn("cs.tyreLoad", cs.tyreLoad, 4); Pr.pn("cs.tyreAngularSpeed", cs.tyreAngularSpeed, 4);
    Pr.pn("cs.tyreSlipRatio", cs.tyreSlipRatio, 4); Pr.pn("cs.tyreNdSlip", cs.tyreNdSlip, 4);
    Pr.pn("cs.probes", cs.probes, 7); Pr.pn("cs.lookAhead", cs.lookAhead, 5);
    Pr.p("cs.stepReward", cs.stepReward); Pr.p("cs.totalReward", cs.totalReward);
}

}  // namespace cpuref
