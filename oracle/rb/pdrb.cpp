// ORACLE / TEST INFRASTRUCTURE -- see pdrb.h header comment.  PARITY UNPINNED (ODE library absent).
#include "pdrb.h"
#include "../cpu_ref/mathsel.h"
#include <cmath>
#include <cstring>
#include <algorithm>

namespace pdrb {

// ---------------------------------------------------------------------------------------------
// small math; evaluation order follows thirdparty/ode/include/ode/odemath.h (dCalcVectorDot3 :213,
// dMultiplyHelper0_331 :317, dMultiplyHelper1_331 :326)
// ---------------------------------------------------------------------------------------------
void mul0_331(float* r, const float* M, const float* v) {
    const float r0 = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
    const float r1 = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
    const float r2 = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
    r[0] = r0; r[1] = r1; r[2] = r2;
}
void mul1_331(float* r, const float* M, const float* v) {
    const float r0 = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
    const float r1 = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
    const float r2 = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
    r[0] = r0; r[1] = r1; r[2] = r2;
}
// res = a * b (3x3)
static void mul0_333(float* res, const float* a, const float* b) {
    float t[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            t[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
    memcpy(res, t, sizeof(t));
}
// res = a * b^T
static void mul2_333(float* res, const float* a, const float* b) {
    float t[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            t[i * 3 + j] = a[i * 3 + 0] * b[j * 3 + 0] + a[i * 3 + 1] * b[j * 3 + 1] + a[i * 3 + 2] * b[j * 3 + 2];
    memcpy(res, t, sizeof(t));
}

void normalize3(float v[3]) {
    const float l = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    if (l > 0.0f) {
        const float s = 1.0f / sqrtf(l);
        v[0] *= s; v[1] *= s; v[2] *= s;
    } else {
        v[0] = 1; v[1] = 0; v[2] = 0;
    }
}
void normalize4(float q[4]) {
    const float l = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (l > 0.0f) {
        const float s = 1.0f / sqrtf(l);
        q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
    } else {
        q[0] = 1; q[1] = 0; q[2] = 0; q[3] = 0;
    }
}

// ODE rotation.cpp dRfromQ
void rFromQ(float R[9], const float q[4]) {
    const float qq1 = 2 * q[1] * q[1];
    const float qq2 = 2 * q[2] * q[2];
    const float qq3 = 2 * q[3] * q[3];
    R[0] = 1 - qq2 - qq3;
    R[1] = 2 * (q[1] * q[2] - q[0] * q[3]);
    R[2] = 2 * (q[1] * q[3] + q[0] * q[2]);
    R[3] = 2 * (q[1] * q[2] + q[0] * q[3]);
    R[4] = 1 - qq1 - qq3;
    R[5] = 2 * (q[2] * q[3] - q[0] * q[1]);
    R[6] = 2 * (q[1] * q[3] - q[0] * q[2]);
    R[7] = 2 * (q[2] * q[3] + q[0] * q[1]);
    R[8] = 1 - qq1 - qq2;
}

// ODE rotation.cpp dQfromR
void qFromR(float q[4], const float R[9]) {
#define RR(i, j) R[(i) * 3 + (j)]
    const float tr = RR(0, 0) + RR(1, 1) + RR(2, 2);
    float s;
    if (tr >= 0) {
        s = sqrtf(tr + 1);
        q[0] = 0.5f * s;
        s = 0.5f * (1.0f / s);
        q[1] = (RR(2, 1) - RR(1, 2)) * s;
        q[2] = (RR(0, 2) - RR(2, 0)) * s;
        q[3] = (RR(1, 0) - RR(0, 1)) * s;
        return;
    }
    int c;
    if (RR(1, 1) > RR(0, 0)) c = (RR(2, 2) > RR(1, 1)) ? 2 : 1;
    else c = (RR(2, 2) > RR(0, 0)) ? 2 : 0;
    if (c == 0) {
        s = sqrtf((RR(0, 0) - (RR(1, 1) + RR(2, 2))) + 1);
        q[1] = 0.5f * s;
        s = 0.5f * (1.0f / s);
        q[2] = (RR(0, 1) + RR(1, 0)) * s;
        q[3] = (RR(2, 0) + RR(0, 2)) * s;
        q[0] = (RR(2, 1) - RR(1, 2)) * s;
    } else if (c == 1) {
        s = sqrtf((RR(1, 1) - (RR(2, 2) + RR(0, 0))) + 1);
        q[2] = 0.5f * s;
        s = 0.5f * (1.0f / s);
        q[3] = (RR(1, 2) + RR(2, 1)) * s;
        q[1] = (RR(0, 1) + RR(1, 0)) * s;
        q[0] = (RR(0, 2) - RR(2, 0)) * s;
    } else {
        s = sqrtf((RR(2, 2) - (RR(0, 0) + RR(1, 1))) + 1);
        q[3] = 0.5f * s;
        s = 0.5f * (1.0f / s);
        q[1] = (RR(2, 0) + RR(0, 2)) * s;
        q[2] = (RR(1, 2) + RR(2, 1)) * s;
        q[0] = (RR(1, 0) - RR(0, 1)) * s;
    }
#undef RR
}

// qa = qb * qc
static void qmul0(float* qa, const float* qb, const float* qc) {
    const float a0 = qb[0] * qc[0] - qb[1] * qc[1] - qb[2] * qc[2] - qb[3] * qc[3];
    const float a1 = qb[0] * qc[1] + qb[1] * qc[0] + qb[2] * qc[3] - qb[3] * qc[2];
    const float a2 = qb[0] * qc[2] + qb[2] * qc[0] + qb[3] * qc[1] - qb[1] * qc[3];
    const float a3 = qb[0] * qc[3] + qb[3] * qc[0] + qb[1] * qc[2] - qb[2] * qc[1];
    qa[0] = a0; qa[1] = a1; qa[2] = a2; qa[3] = a3;
}
// qa = inv(qb) * qc
static void qmul1(float* qa, const float* qb, const float* qc) {
    const float a0 = qb[0] * qc[0] + qb[1] * qc[1] + qb[2] * qc[2] + qb[3] * qc[3];
    const float a1 = qb[0] * qc[1] - qb[1] * qc[0] - qb[2] * qc[3] + qb[3] * qc[2];
    const float a2 = qb[0] * qc[2] - qb[2] * qc[0] - qb[3] * qc[1] + qb[1] * qc[3];
    const float a3 = qb[0] * qc[3] - qb[3] * qc[0] - qb[1] * qc[2] + qb[2] * qc[1];
    qa[0] = a0; qa[1] = a1; qa[2] = a2; qa[3] = a3;
}
// qa = qb * inv(qc)
static void qmul2(float* qa, const float* qb, const float* qc) {
    const float a0 = qb[0] * qc[0] + qb[1] * qc[1] + qb[2] * qc[2] + qb[3] * qc[3];
    const float a1 = -qb[0] * qc[1] + qb[1] * qc[0] - qb[2] * qc[3] + qb[3] * qc[2];
    const float a2 = -qb[0] * qc[2] + qb[2] * qc[0] - qb[3] * qc[1] + qb[1] * qc[3];
    const float a3 = -qb[0] * qc[3] + qb[3] * qc[0] - qb[1] * qc[2] + qb[2] * qc[1];
    qa[0] = a0; qa[1] = a1; qa[2] = a2; qa[3] = a3;
}

// ODE odemath.cpp dPlaneSpace
void planeSpace(const float n[3], float p[3], float q[3]) {
    if (fabsf(n[2]) > 0.70710678118654752440f) {
        const float a = n[1] * n[1] + n[2] * n[2];
        const float k = 1.0f / sqrtf(a);
        p[0] = 0; p[1] = -n[2] * k; p[2] = n[1] * k;
        q[0] = a * k; q[1] = -n[0] * p[2]; q[2] = n[0] * p[1];
    } else {
        const float a = n[0] * n[0] + n[1] * n[1];
        const float k = 1.0f / sqrtf(a);
        p[0] = -n[1] * k; p[1] = n[0] * k; p[2] = 0;
        q[0] = -n[2] * p[1]; q[1] = n[2] * p[0]; q[2] = a * k;
    }
}

// ---------------------------------------------------------------------------------------------
// Body
// ---------------------------------------------------------------------------------------------
void Body::setMassBoxTotal(float m, float lx, float ly, float lz) {
    // dMassSetBoxTotal = dMassSetBox(density 1) then dMassAdjust (RigidBodyODE.cpp:64-70)
    const float M = lx * ly * lz * 1.0f;
    float i0 = M / 12.0f * (ly * ly + lz * lz);
    float i1 = M / 12.0f * (lx * lx + lz * lz);
    float i2 = M / 12.0f * (lx * lx + ly * ly);
    const float scale = m / M;
    i0 *= scale; i1 *= scale; i2 *= scale;
    mass = m;
    invMass = 1.0f / m;
    for (int k = 0; k < 9; ++k) { I[k] = 0; invI[k] = 0; }
    I[0] = i0; I[4] = i1; I[8] = i2;
    invI[0] = 1.0f / i0; invI[4] = 1.0f / i1; invI[8] = 1.0f / i2;
}

void Body::setRotation(const float Rin[9]) {
    // dBodySetRotation: copy, dOrthogonalizeR, q = dQfromR(R_in), normalise q
    float m[9];
    memcpy(m, Rin, sizeof(m));
    const float n0 = m[0] * m[0] + m[1] * m[1] + m[2] * m[2];
    if (n0 != 1.0f) normalize3(m);
    const float proj = m[0] * m[3] + m[1] * m[4] + m[2] * m[5];
    if (proj != 0.0f) { m[3] -= proj * m[0]; m[4] -= proj * m[1]; m[5] -= proj * m[2]; }
    const float n1 = m[3] * m[3] + m[4] * m[4] + m[5] * m[5];
    if (n1 != 1.0f) normalize3(m + 3);
    cross3(m + 6, m, m + 3);
    memcpy(R, m, sizeof(m));
    qFromR(q, Rin);
    normalize4(q);
}

void Body::relPointPos(const float p[3], float out[3]) const {
    float t[3];
    mul0_331(t, R, p);
    out[0] = t[0] + pos[0]; out[1] = t[1] + pos[1]; out[2] = t[2] + pos[2];
}
void Body::posRelPoint(const float p[3], float out[3]) const {
    const float d[3] = {p[0] - pos[0], p[1] - pos[1], p[2] - pos[2]};
    mul1_331(out, R, d);
}
void Body::vectorToWorld(const float v[3], float out[3]) const { mul0_331(out, R, v); }
void Body::vectorFromWorld(const float v[3], float out[3]) const { mul1_331(out, R, v); }
void Body::relPointVel(const float p[3], float out[3]) const {
    float w[3], c[3];
    mul0_331(w, R, p);
    cross3(c, avel, w);
    out[0] = lvel[0] + c[0]; out[1] = lvel[1] + c[1]; out[2] = lvel[2] + c[2];
}
void Body::pointVel(const float p[3], float out[3]) const {
    const float d[3] = {p[0] - pos[0], p[1] - pos[1], p[2] - pos[2]};
    float c[3];
    cross3(c, avel, d);
    out[0] = lvel[0] + c[0]; out[1] = lvel[1] + c[1]; out[2] = lvel[2] + c[2];
}
void Body::addForceAtPos(const float f[3], const float p[3]) {
    facc[0] += f[0]; facc[1] += f[1]; facc[2] += f[2];
    const float d[3] = {p[0] - pos[0], p[1] - pos[1], p[2] - pos[2]};
    float c[3];
    cross3(c, d, f);
    tacc[0] += c[0]; tacc[1] += c[1]; tacc[2] += c[2];
}
void Body::addForceAtRelPos(const float f[3], const float p[3]) {
    float w[3], c[3];
    mul0_331(w, R, p);
    facc[0] += f[0]; facc[1] += f[1]; facc[2] += f[2];
    cross3(c, w, f);
    tacc[0] += c[0]; tacc[1] += c[1]; tacc[2] += c[2];
}
void Body::addRelForceAtRelPos(const float fl[3], const float p[3]) {
    float f[3], w[3], c[3];
    mul0_331(f, R, fl);
    mul0_331(w, R, p);
    facc[0] += f[0]; facc[1] += f[1]; facc[2] += f[2];
    cross3(c, w, f);
    tacc[0] += c[0]; tacc[1] += c[1]; tacc[2] += c[2];
}
void Body::addRelForceAtPos(const float fl[3], const float p[3]) {
    float f[3];
    mul0_331(f, R, fl);
    addForceAtPos(f, p);
}
void Body::addTorque(const float t[3]) { tacc[0] += t[0]; tacc[1] += t[1]; tacc[2] += t[2]; }
void Body::addRelTorque(const float tl[3]) {
    float t[3];
    mul0_331(t, R, tl);
    tacc[0] += t[0]; tacc[1] += t[1]; tacc[2] += t[2];
}
void Body::stop() {
    for (int k = 0; k < 3; ++k) { lvel[k] = 0; avel[k] = 0; facc[k] = 0; tacc[k] = 0; }
}

// ---------------------------------------------------------------------------------------------
// Joint creation (ODE setters take WORLD coordinates; JointODE.cpp:21-89)
// ---------------------------------------------------------------------------------------------
int World::createFixed(int b0, int b1) {
    Joint j;
    j.type = JT_FIXED; j.b0 = b0; j.b1 = b1; j.erp = erp; j.cfm = cfm;
    const Body& A = bodies[b0];
    const Body& B = bodies[b1];
    // dJointSetFixed: offset = R0^T (p0 - p1); qrel = inv(q0) q1
    const float ofs[3] = {A.pos[0] - B.pos[0], A.pos[1] - B.pos[1], A.pos[2] - B.pos[2]};
    mul1_331(j.offset, A.R, ofs);
    qmul1(j.qrel, A.q, B.q);
    joints.push_back(j);
    orderDirty = true;
    return (int)joints.size() - 1;
}
int World::createBall(int b0, int b1, const float aw[3]) {
    Joint j;
    j.type = JT_BALL; j.b0 = b0; j.b1 = b1; j.erp = erp; j.cfm = cfm;
    bodies[b0].posRelPoint(aw, j.anchor1);
    bodies[b1].posRelPoint(aw, j.anchor2);
    joints.push_back(j);
    orderDirty = true;
    return (int)joints.size() - 1;
}
int World::createSlider(int b0, int b1, const float axisWorld[3]) {
    Joint j;
    j.type = JT_SLIDER; j.b0 = b0; j.b1 = b1; j.erp = erp; j.cfm = cfm;
    const Body& A = bodies[b0];
    const Body& B = bodies[b1];
    float a[3] = {axisWorld[0], axisWorld[1], axisWorld[2]};
    normalize3(a);
    mul1_331(j.axis1, A.R, a);
    // computeOffset: offset = R1^T (p0 - p1); computeInitialRelativeRotation: qrel = inv(q0) q1
    const float c[3] = {A.pos[0] - B.pos[0], A.pos[1] - B.pos[1], A.pos[2] - B.pos[2]};
    mul1_331(j.offset, B.R, c);
    qmul1(j.qrel, A.q, B.q);
    joints.push_back(j);
    orderDirty = true;
    return (int)joints.size() - 1;
}
int World::createDBall(int b0, int b1, const float a1w[3], const float a2w[3]) {
    Joint j;
    j.type = JT_DBALL; j.b0 = b0; j.b1 = b1; j.erp = erp; j.cfm = cfm;
    joints.push_back(j);
    const int id = (int)joints.size() - 1;
    dballSetAnchor1(id, a1w);
    dballSetAnchor2(id, a2w);
    orderDirty = true;
    return id;
}
void World::dballUpdateTargetDistance(int id) {
    Joint& j = joints[id];
    float g1[3], g2[3];
    bodies[j.b0].relPointPos(j.anchor1, g1);
    bodies[j.b1].relPointPos(j.anchor2, g2);
    const float d[3] = {g1[0] - g2[0], g1[1] - g2[1], g1[2] - g2[2]};
    j.targetDistance = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
}
void World::dballSetAnchor1(int id, const float w[3]) {
    bodies[joints[id].b0].posRelPoint(w, joints[id].anchor1);
    dballUpdateTargetDistance(id);
}
void World::dballSetAnchor2(int id, const float w[3]) {
    bodies[joints[id].b1].posRelPoint(w, joints[id].anchor2);
    dballUpdateTargetDistance(id);
}

// ---------------------------------------------------------------------------------------------
// Island traversal order (ODE util.cpp dxProcessIslands): bodies are visited from the head of the
// world's body list (most recently created first); each body's joint list has the most recently
// attached joint first; depth-first with an explicit stack.  All joints used here are fully
// unbounded (nub == m), so the stepper's unbounded/mixed/LCP regrouping leaves this order intact.
// ---------------------------------------------------------------------------------------------
void World::buildOrder() {
    const int nb = (int)bodies.size();
    const int nj = (int)joints.size();
    std::vector<std::vector<int>> bj(nb);
    for (int j = 0; j < nj; ++j) {  // attach order == creation order here
        bj[joints[j].b0].insert(bj[joints[j].b0].begin(), j);
        if (joints[j].b1 >= 0) bj[joints[j].b1].insert(bj[joints[j].b1].begin(), j);
    }
    std::vector<char> btag(nb, 0), jtag(nj, 0);
    jointOrder.clear();
    std::vector<int> stack;
    for (int bb = nb - 1; bb >= 0; --bb) {
        if (btag[bb]) continue;
        btag[bb] = 1;
        int b = bb;
        stack.clear();
        while (true) {
            for (int j : bj[b]) {
                if (jtag[j]) continue;
                jtag[j] = 1;
                jointOrder.push_back(j);
                const int other = (joints[j].b0 == b) ? joints[j].b1 : joints[j].b0;
                if (other >= 0 && !btag[other]) { btag[other] = 1; stack.push_back(other); }
            }
            if (stack.empty()) break;
            b = stack.back();
            stack.pop_back();
        }
    }
    orderDirty = false;
}

// ---------------------------------------------------------------------------------------------
// joint rows.  Row layout: J[12] = {J1l(3), J1a(3), J2l(3), J2a(3)}, c, cfm.
// ---------------------------------------------------------------------------------------------
struct Row { float J[12]; float c; float cfm; };

static void setFixedOrientation(const Body& A, const Body& B, const Joint& j, float k, Row* rows) {
    // ODE joints/joint.cpp setFixedOrientation: J1a = I, J2a = -I, c = 2 k R0 * vec(qerr)
    for (int r = 0; r < 3; ++r) {
        for (int t = 0; t < 12; ++t) rows[r].J[t] = 0;
        rows[r].J[3 + r] = 1.0f;
        rows[r].J[9 + r] = -1.0f;
    }
    float qq[4], qerr[4], e[3];
    qmul1(qq, A.q, B.q);
    qmul2(qerr, qq, j.qrel);
    if (qerr[0] < 0) { qerr[1] = -qerr[1]; qerr[2] = -qerr[2]; qerr[3] = -qerr[3]; }
    mul0_331(e, A.R, qerr + 1);
    rows[0].c = 2 * k * e[0];
    rows[1].c = 2 * k * e[1];
    rows[2].c = 2 * k * e[2];
}

static int jointRows(const World& w, const Joint& j, float fps, Row* rows) {
    const Body& A = w.bodies[j.b0];
    const Body& B = w.bodies[j.b1];
    const int m = j.rows();
    for (int r = 0; r < m; ++r) {
        for (int t = 0; t < 12; ++t) rows[r].J[t] = 0;
        rows[r].c = 0;
        rows[r].cfm = w.cfm;
    }
    switch (j.type) {
    case JT_BALL: {
        // ODE joints/ball.cpp + setBall: J1l = I, J1a = -[a1]x, J2l = -I, J2a = +[a2]x
        float a1[3], a2[3];
        mul0_331(a1, A.R, j.anchor1);
        mul0_331(a2, B.R, j.anchor2);
        for (int r = 0; r < 3; ++r) { rows[r].J[r] = 1.0f; rows[r].J[6 + r] = -1.0f; rows[r].cfm = j.cfm; }
        // dSetCrossMatrixMinus(J1a, a1) (odemath.h:288)
        rows[0].J[3 + 1] = +a1[2]; rows[0].J[3 + 2] = -a1[1];
        rows[1].J[3 + 0] = -a1[2]; rows[1].J[3 + 2] = +a1[0];
        rows[2].J[3 + 0] = +a1[1]; rows[2].J[3 + 1] = -a1[0];
        // dSetCrossMatrixPlus(J2a, a2) (odemath.h:277)
        rows[0].J[9 + 1] = -a2[2]; rows[0].J[9 + 2] = +a2[1];
        rows[1].J[9 + 0] = +a2[2]; rows[1].J[9 + 2] = -a2[0];
        rows[2].J[9 + 0] = -a2[1]; rows[2].J[9 + 1] = +a2[0];
        const float k = fps * j.erp;
        for (int r = 0; r < 3; ++r) rows[r].c = k * (a2[r] + B.pos[r] - a1[r] - A.pos[r]);
        break;
    }
    case JT_FIXED: {
        // ODE joints/fixed.cpp: rows 0-2 position, rows 3-5 orientation
        float ofs[3];
        mul0_331(ofs, A.R, j.offset);
        for (int r = 0; r < 3; ++r) { rows[r].J[r] = 1.0f; rows[r].J[6 + r] = -1.0f; rows[r].cfm = j.cfm; }
        // dSetCrossMatrixPlus(J1a, ofs)
        rows[0].J[3 + 1] = -ofs[2]; rows[0].J[3 + 2] = +ofs[1];
        rows[1].J[3 + 0] = +ofs[2]; rows[1].J[3 + 2] = -ofs[0];
        rows[2].J[3 + 0] = -ofs[1]; rows[2].J[3 + 1] = +ofs[0];
        const float k = fps * j.erp;
        for (int r = 0; r < 3; ++r) rows[r].c = k * (B.pos[r] - A.pos[r] + ofs[r]);
        setFixedOrientation(A, B, j, fps * w.erp, rows + 3);
        for (int r = 3; r < 6; ++r) rows[r].cfm = w.cfm;
        break;
    }
    case JT_SLIDER: {
        // ODE joints/slider.cpp getInfo2 (no limit / motor row: no stops, fmax = 0)
        float c[3] = {B.pos[0] - A.pos[0], B.pos[1] - A.pos[1], B.pos[2] - A.pos[2]};
        const float k = fps * w.erp;
        setFixedOrientation(A, B, j, k, rows);
        float ax1[3], p[3], q[3], tmp[3];
        mul0_331(ax1, A.R, j.axis1);
        planeSpace(ax1, p, q);
        cross3(tmp, c, p);
        for (int t = 0; t < 3; ++t) { tmp[t] *= 0.5f; rows[3].J[3 + t] = tmp[t]; rows[3].J[9 + t] = tmp[t]; }
        cross3(tmp, c, q);
        for (int t = 0; t < 3; ++t) { tmp[t] *= 0.5f; rows[4].J[3 + t] = tmp[t]; rows[4].J[9 + t] = tmp[t]; }
        for (int t = 0; t < 3; ++t) {
            rows[3].J[6 + t] = -p[t]; rows[4].J[6 + t] = -q[t];
            rows[3].J[t] = p[t];      rows[4].J[t] = q[t];
        }
        float ofs[3];
        mul0_331(ofs, B.R, j.offset);
        for (int t = 0; t < 3; ++t) c[t] += ofs[t];
        rows[3].c = k * dot3(p, c);
        rows[4].c = k * dot3(q, c);
        break;
    }
    case JT_DBALL: {
        // ODE joints/dball.cpp getInfo2
        float g1[3], g2[3], q[3];
        A.relPointPos(j.anchor1, g1);
        B.relPointPos(j.anchor2, g2);
        q[0] = g1[0] - g2[0]; q[1] = g1[1] - g2[1]; q[2] = g1[2] - g2[2];
        const float dist = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
        if (dist < 1e-7f) {
            // too small: direction from anchor velocity difference, else arbitrary
            float v1[3], v2[3];
            A.relPointVel(j.anchor1, v1);
            B.relPointVel(j.anchor2, v2);
            q[0] = v1[0] - v2[0]; q[1] = v1[1] - v2[1]; q[2] = v1[2] - v2[2];
            if (sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]) < 1e-7f) { q[0] = 1; q[1] = 0; q[2] = 0; }
        }
        normalize3(q);
        float r1[3], r2[3], t[3];
        mul0_331(r1, A.R, j.anchor1);
        mul0_331(r2, B.R, j.anchor2);
        for (int k2 = 0; k2 < 3; ++k2) { rows[0].J[k2] = q[k2]; rows[0].J[6 + k2] = -q[k2]; }
        cross3(t, r1, q);  // (-[r1]x)^T q = r1 x q
        rows[0].J[3] = t[0]; rows[0].J[4] = t[1]; rows[0].J[5] = t[2];
        cross3(t, q, r2);  // ([r2]x)^T q = -(r2 x q) = q x r2
        rows[0].J[9] = t[0]; rows[0].J[10] = t[1]; rows[0].J[11] = t[2];
        rows[0].cfm = j.cfm;
        rows[0].c = (fps * j.erp) * (j.targetDistance - dist);
        break;
    }
    }
    return m;
}

// ---------------------------------------------------------------------------------------------
// dWorldStep (ODE step.cpp dxStepIsland), all rows unbounded => one SPD factor-solve.
// Canonical factorisation (project choice): right-looking LDL^T with fmaf, column-oriented
// substitutions -- each element sees its updates in k order, independent of any parallel split.
// ---------------------------------------------------------------------------------------------
static inline float sinc_ode(float x) {
    if (fabsf(x) < 1.0e-4f) return 1.0f - x * x * 0.166666666666666666667f;
    return m_sinf(x) / x;
}

// ---------------------------------------------------------------------------------------------
// Contact joints (ODE joints/contact.cpp getInfo1/getInfo2) and the mixed LCP they turn dWorldStep into (ODE lcp.cpp).
//
// Rows of contact c (3: mu > 0 and finite): normal n (lo 0, hi inf, rhs = min(fps*erp*depth, maxCorrectingVel), raised to
// bounce * approach speed when dContactBounce; cfm = soft_cfm), then the two friction directions of dPlaneSpace(n) with
// lo/hi = -/+ mu and findex = the normal row (dContactApprox1: Dantzig fixes the friction limits to mu * |lambda_n| when it
// reaches the first friction row, from the solution of all the other rows WITHOUT friction; they are not updated afterwards).
//
// Solution method (this project's; ODE uses Dantzig's pivoting on the whole matrix -- same unique solution of the same two
// LCPs, different rounding): every contact is on ONE body (the chassis), so with the unbounded rows eliminated first (which is
// also what Dantzig does: nub rows first) the contact rows see a 6x6 "effective inverse mass" of that body,
//     K = Minv_b - G^T Auu^-1 G,   G = the body's block of J_u Minv,
// S = Jc K Jc^T + cfm/h, rhs_c = c/h - Jc (tmp1_b + G^T lambda0).  S is six columns and a diagonal: with K = Lk diag(dk) Lk^T and
// Jh = Jc Lk, S = Jh diag(dk) Jh^T + diag(cfm/h) is never formed.  Given the body's answer tot = dk (.) Jh^T x, every row stands
// alone, x_r = clamp((rhs_r - Jh_r . tot) / (cfm_r/h), lo_r, hi_r): the box LCP is a piecewise-linear equation in six unknowns,
// the gradient of a strictly convex function, solved by Newton's method with a bisection line search (solveContacts below).
// Stage 1: normal rows only (the friction rows are Dantzig's "don't care" rows, x = 0); stage 2: all rows with the limits fixed.
// Then lambda_u = lambda0 - (Auu^-1 G) (Jc^T lambda_c).
// ---------------------------------------------------------------------------------------------
static inline float dot6c(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5]; }
enum { ST_FREE = 0, ST_LO = 1, ST_HI = 2, ST_FIXED = 3, ST_OFF = 4 };

// A = L diag(d) L^T of a symmetric positive definite 6x6 (lower triangle of A read; L unit lower, row-major)
static inline void ldl6(const float* A, float* L, float* d, float* dinv) {
    for (int j = 0; j < 6; ++j) {
        float t[6];
        float dj = A[j * 6 + j];
        for (int k = 0; k < j; ++k) { t[k] = L[j * 6 + k] * d[k]; dj = fmaf(-L[j * 6 + k], t[k], dj); }
        d[j] = dj; dinv[j] = 1.0f / dj;
        for (int i = j + 1; i < 6; ++i) {
            float vv = A[i * 6 + j];
            for (int k = 0; k < j; ++k) vv = fmaf(-L[i * 6 + k], t[k], vv);
            L[i * 6 + j] = vv * dinv[j];
        }
    }
}

static void solveContacts(World& w, float fps, int m, const int* rb0, const int* rb1, const float* JinvM, const float* L, const float* dinv,
                          const float* invIwB, const float* tmp1B, float* lambda, float* contactForce) {
    const int nc = (int)w.contacts.size();
    const int nr = 3 * nc;
    const int cb = w.contactBody;
    const Body& B = w.bodies[cb];
    std::vector<float> Jc(nr * 6), cv(nr), cfmc(nr), lo(nr), hi(nr), mu(nr);
    for (int c = 0; c < nc; ++c) {
        const ContactJoint& cj = w.contacts[c];
        const bool box = cj.kind == 1;
        const float muC = box ? 0.1f : 0.25f, bounce = box ? 0.0f : 0.01f, softCfm = box ? 0.000952380942f : 0.0001f;
        const float erpN = box ? 0.714285731f : w.erp;
        const float n[3] = {cj.normal[0], cj.normal[1], cj.normal[2]};
        const float c1[3] = {cj.pos[0] - B.pos[0], cj.pos[1] - B.pos[1], cj.pos[2] - B.pos[2]};
        float t1[3], t2[3];
        planeSpace(n, t1, t2);
        const float* dir[3] = {n, t1, t2};
        for (int k = 0; k < 3; ++k) {
            float* J = &Jc[(3 * c + k) * 6];
            J[0] = dir[k][0]; J[1] = dir[k][1]; J[2] = dir[k][2];
            cross3(J + 3, c1, dir[k]);
        }
        const float kk = fps * erpN;
        float depth = cj.depth - w.contactSurfaceLayer;
        if (depth < 0.0f) depth = 0.0f;
        const float pushout = kk * depth;
        float cN = pushout > w.contactMaxCorrectingVel ? w.contactMaxCorrectingVel : pushout;
        const float* Jn = &Jc[(3 * c) * 6];
        const float outgoing = dot3(Jn + 3, B.avel) + dot3(n, B.lvel);
        const float negOut = -outgoing;
        if (negOut > 0.0f) {   // bounce_vel = 0 (memzero'd dContact)
            const float newc = bounce * negOut;
            if (newc > cN) cN = newc;
        }
        cv[3 * c] = cN; cv[3 * c + 1] = 0.0f; cv[3 * c + 2] = 0.0f;
        cfmc[3 * c] = softCfm; cfmc[3 * c + 1] = w.cfm; cfmc[3 * c + 2] = w.cfm;
        lo[3 * c] = 0.0f; hi[3 * c] = 3.0e38f;
        mu[3 * c] = 0.0f; mu[3 * c + 1] = muC; mu[3 * c + 2] = muC;
    }
    // G = the contact body's block of J_u Minv; W = Auu^-1 G (six more right-hand sides through the factor)
    std::vector<float> G(m * 6 + 1, 0.0f), W(m * 6 + 1, 0.0f);
    for (int i = 0; i < m; ++i) {
        const float* src = (rb0[i] == cb) ? &JinvM[i * 12] : (rb1[i] == cb) ? &JinvM[i * 12 + 6] : nullptr;
        for (int a = 0; a < 6; ++a) { G[i * 6 + a] = src ? src[a] : 0.0f; W[i * 6 + a] = G[i * 6 + a]; }
    }
    for (int a = 0; a < 6; ++a) {
        for (int k = 0; k < m; ++k)
            for (int i = k + 1; i < m; ++i) W[i * 6 + a] = fmaf(-L[i * m + k], W[k * 6 + a], W[i * 6 + a]);
        for (int k = 0; k < m; ++k) W[k * 6 + a] *= dinv[k];
        for (int k = m - 1; k >= 0; --k)
            for (int i = 0; i < k; ++i) W[i * 6 + a] = fmaf(-L[k * m + i], W[k * 6 + a], W[i * 6 + a]);
    }
    float K[36], u[6];
    for (int a = 0; a < 6; ++a) {
        for (int b = 0; b < 6; ++b) {
            float minv = 0.0f;
            if (a < 3 && b < 3) minv = (a == b) ? B.invMass : 0.0f;
            else if (a >= 3 && b >= 3) minv = invIwB[(a - 3) * 3 + (b - 3)];
            float acc = 0.0f;
            for (int i = 0; i < m; ++i) acc += G[i * 6 + a] * W[i * 6 + b];
            K[a * 6 + b] = minv - acc;
        }
        float acc = 0.0f;
        for (int i = 0; i < m; ++i) acc += G[i * 6 + a] * lambda[i];
        u[a] = tmp1B[a] + acc;
    }
    // ---- the contact rows in the body's 6-space ----
    // K = Lk diag(dk) Lk^T (lower triangle of K as computed); Jh_r = J_r Lk; S = Jh diag(dk) Jh^T + diag(dd), dd = cfm/h.
    float Lk[36], dk[6], dkinv[6];
    ldl6(K, Lk, dk, dkinv);
    std::vector<float> Jh(nr * 6), dd(nr), idd(nr), bt(nr);
    for (int r = 0; r < nr; ++r) {
        const float* J = &Jc[r * 6];
        for (int a2 = 0; a2 < 6; ++a2) {
            float acc = J[a2];
            for (int b2 = a2 + 1; b2 < 6; ++b2) acc = fmaf(J[b2], Lk[b2 * 6 + a2], acc);
            Jh[r * 6 + a2] = acc;
        }
        dd[r] = cfmc[r] * fps;
        idd[r] = 1.0f / dd[r];
        float rr = cv[r] * fps;
        rr -= dot6c(J, u);
        bt[r] = rr;
    }
    std::vector<int> state(nr);
    std::vector<float> x(nr, 0.0f);
    int iterations = 0;
#ifndef PDRB_LCP_MAX_ITERATIONS
#define PDRB_LCP_MAX_ITERATIONS 64   /* per stage; = the kernel's.  Measured on the playground: median 3 rounds per solve (both stages), 99 % within 7, maximum 12 */
#endif
    // one quantity summed over the rows: contact c's three rows first ((t0 + t1) + t2, a row outside the mask gives 0), then
    // the contacts as the leaves of a balanced binary tree over 64 slots (the kernel's lane = contact butterfly)
    float leaf[64];
    auto tree = [&]() { for (int n2 = 32; n2 >= 1; n2 >>= 1) for (int j = 0; j < n2; ++j) leaf[j] = leaf[2 * j] + leaf[2 * j + 1]; return leaf[0]; };
    auto sum = [&](auto term) {
        for (int c = 0; c < 64; ++c) { leaf[c] = 0.0f; if (c < nc) leaf[c] = (term(3 * c) + term(3 * c + 1)) + term(3 * c + 2); }
        return tree();
    };
    // Given the body's answer `tot` (6 numbers), every row stands alone: x_r = clamp((bt_r - Jh_r . tot) / dd_r, lo_r, hi_r), and
    // tot = dk (.) sum_r Jh_r^T x_r closes the loop -- a piecewise-linear equation in six unknowns, the gradient of a strictly
    // convex function.  Newton's method on it (one step = the rows sorted into free / at a bound by the clamp at the current
    // point, the free ones solved exactly through the 6x6 system), with the step length by bisection on the directional
    // derivative when the full step overshoots: the rounds descend, there is nothing to cycle through.
    float totc[6] = {0, 0, 0, 0, 0, 0};
    std::vector<float> ar(nr, 0.0f), cr(nr, 0.0f);   // a row's Jh_r . (step) and Jh_r . (current point)
    // the rows at totc + t step: where the clamp puts them; commit = take it over, else count the rows that would change
    auto place = [&](float t, bool commit) {
        int changes = 0;
        for (int r = 0; r < nr; ++r) {
            if (state[r] == ST_OFF || state[r] == ST_FIXED) { if (commit) x[r] = 0.0f; continue; }
            const float xt = (bt[r] - fmaf(t, ar[r], cr[r])) * idd[r];
            const int ns = xt < lo[r] ? ST_LO : xt > hi[r] ? ST_HI : ST_FREE;
            if (ns != state[r]) ++changes;
            if (commit) { state[r] = ns; x[r] = ns == ST_FREE ? xt : ns == ST_LO ? lo[r] : hi[r]; }
        }
        return changes;
    };
    auto newton = [&]() {
        for (int r = 0; r < nr; ++r) { ar[r] = 0.0f; cr[r] = dot6c(&Jh[r * 6], totc); }
        place(0.0f, true);
        for (int it = 0; it < PDRB_LCP_MAX_ITERATIONS; ++it) {
            ++iterations;
            // the masks of the sums: a free row's 1/dd (else 0), a bounded row's x (else 0)
            std::vector<float> fi(nr), xb(nr);
            for (int r = 0; r < nr; ++r) { fi[r] = state[r] == ST_FREE ? idd[r] : 0.0f; xb[r] = (state[r] == ST_LO || state[r] == ST_HI) ? x[r] : 0.0f; }
            // v = sum over the rows at a bound of Jh_r x_r;  N = sum over the free rows of Jh_r^T Jh_r / dd_r;  gb = sum over the free rows of Jh_r bt_r / dd_r
            float v[6], N[36], gb[6];
            for (int a2 = 0; a2 < 6; ++a2) {
                v[a2] = sum([&](int r) { return Jh[r * 6 + a2] * xb[r]; });
                gb[a2] = sum([&](int r) { return (Jh[r * 6 + a2] * fi[r]) * bt[r]; });
                for (int b2 = 0; b2 <= a2; ++b2) {
                    N[a2 * 6 + b2] = sum([&](int r) { return (Jh[r * 6 + a2] * fi[r]) * Jh[r * 6 + b2]; });
                    N[b2 * 6 + a2] = N[a2 * 6 + b2];
                }
            }
            // (diag(1/dk) + N) wh = gb - N vb, vb = dk v: the free rows' pull on the body
            float vb[6], g[6], M[36], Lm[36], dm[6], dminv[6], wh[6], totn[6], dl[6];
            for (int a2 = 0; a2 < 6; ++a2) vb[a2] = dk[a2] * v[a2];
            for (int a2 = 0; a2 < 6; ++a2) {
                float acc = gb[a2];
                for (int b2 = 0; b2 < 6; ++b2) acc = fmaf(-N[a2 * 6 + b2], vb[b2], acc);
                g[a2] = acc;
                for (int b2 = 0; b2 < 6; ++b2) M[a2 * 6 + b2] = N[a2 * 6 + b2];
                M[a2 * 6 + a2] = N[a2 * 6 + a2] + dkinv[a2];
            }
            ldl6(M, Lm, dm, dminv);
            for (int i = 0; i < 6; ++i) { float acc = g[i]; for (int k = 0; k < i; ++k) acc = fmaf(-Lm[i * 6 + k], wh[k], acc); wh[i] = acc; }
            for (int i = 0; i < 6; ++i) wh[i] *= dminv[i];
            for (int i = 5; i >= 0; --i) { float acc = wh[i]; for (int k = i + 1; k < 6; ++k) acc = fmaf(-Lm[k * 6 + i], wh[k], acc); wh[i] = acc; }
            for (int a2 = 0; a2 < 6; ++a2) { totn[a2] = vb[a2] + wh[a2]; dl[a2] = totn[a2] - totc[a2]; }
            for (int r = 0; r < nr; ++r) { ar[r] = dot6c(&Jh[r * 6], dl); cr[r] = dot6c(&Jh[r * 6], totc); }
            // the full step leaves every row where it is: that is the solution
            if (place(1.0f, false) == 0) { place(1.0f, true); for (int a2 = 0; a2 < 6; ++a2) totc[a2] = totn[a2]; break; }
            // along totc + t dl: g(t) = dl . (tot(t) / dk - sum_r Jh_r^T clamp_r(t)), increasing in t, negative at 0
            float e0 = 0.0f, e1 = 0.0f;
            for (int a2 = 0; a2 < 6; ++a2) { const float w6 = dl[a2] * dkinv[a2]; e0 = fmaf(w6, totc[a2], e0); e1 = fmaf(w6, dl[a2], e1); }
            auto slope = [&](float t) {
                const float s2 = sum([&](int r) {
                    if (state[r] == ST_OFF || state[r] == ST_FIXED) return 0.0f;
                    float xt = (bt[r] - fmaf(t, ar[r], cr[r])) * idd[r];
                    xt = xt < lo[r] ? lo[r] : xt > hi[r] ? hi[r] : xt;
                    return xt * ar[r];
                });
                return fmaf(t, e1, e0) - s2;
            };
            float t = 1.0f;
            if (slope(1.0f) > 0.0f) {
                float tl = 0.0f, th = 1.0f;
                for (int k = 0; k < 10; ++k) { const float tm = 0.5f * (tl + th); if (slope(tm) > 0.0f) th = tm; else tl = tm; }
                t = tl;
            }
            if (t == 0.0f) break;   // no descent left at this resolution
            if (t == 1.0f) { for (int a2 = 0; a2 < 6; ++a2) totc[a2] = totn[a2]; }
            else { for (int a2 = 0; a2 < 6; ++a2) totc[a2] = fmaf(t, dl[a2], totc[a2]); }
            place(t, true);
        }
    };
    for (int r = 0; r < nr; ++r) state[r] = (r % 3 == 0) ? ST_FREE : ST_OFF;
    newton();
    for (int r = 0; r < nr; ++r) {
        if (r % 3 == 0) continue;
        const float h2 = fabsf(mu[r] * x[r - r % 3]);
        hi[r] = h2; lo[r] = -h2;
        state[r] = (h2 > 0.0f) ? ST_FREE : ST_FIXED;
    }
    newton();
    // back to the unbounded rows and the body
    float yv[6];
    for (int a = 0; a < 6; ++a) {
        const float acc = sum([&](int r) { return Jc[r * 6 + a] * x[r]; });
        yv[a] = acc;
        contactForce[a] = acc;
    }
    for (int i = 0; i < m; ++i) lambda[i] = lambda[i] - dot6c(&W[i * 6], yv);
    w.lastContactLambda = x; w.lastContactLo = lo; w.lastContactHi = hi; w.lastLcpIterations = iterations;
}

void World::step(float h) {
    if (orderDirty) buildOrder();
    const int nb = (int)bodies.size();
    const float fps = 1.0f / h;

    static thread_local std::vector<float> invIw; invIw.assign(nb * 9, float());
    for (int bi = 0; bi < nb; ++bi) {
        Body& b = bodies[bi];
        float tmp[9];
        float* iw = &invIw[bi * 9];
        mul2_333(tmp, b.invI, b.R);
        mul0_333(iw, b.R, tmp);
        // gyroscopic torque, implicit form (ODE >= 0.13 step.cpp, "Stabilizing Gyroscopic Forces")
        {
            float I[9], L[3];
            mul2_333(tmp, b.I, b.R);
            mul0_333(I, b.R, tmp);
            mul0_331(L, I, b.avel);
            float It[9] = {0, +L[2], -L[1], -L[2], 0, +L[0], +L[1], -L[0], 0};  // dSetCrossMatrixMinus
            for (int k = 0; k < 9; ++k) It[k] = It[k] * h + I[k];
            L[0] *= fps; L[1] *= fps; L[2] *= fps;
            // dInvertMatrix3 (odemath.h:463-503)
            const float det = It[0] * (It[4] * It[8] - It[7] * It[5]) - It[1] * (It[3] * It[8] - It[6] * It[5]) +
                              It[2] * (It[3] * It[7] - It[6] * It[4]);
            if (det != 0.0f) {
                const float dr = 1.0f / det;
                float inv[9];
                inv[0] = (It[4] * It[8] - It[5] * It[7]) * dr;
                inv[1] = (It[7] * It[2] - It[1] * It[8]) * dr;
                inv[2] = (It[1] * It[5] - It[4] * It[2]) * dr;
                inv[3] = (It[5] * It[6] - It[3] * It[8]) * dr;
                inv[4] = (It[0] * It[8] - It[6] * It[2]) * dr;
                inv[5] = (It[3] * It[2] - It[0] * It[5]) * dr;
                inv[6] = (It[3] * It[7] - It[6] * It[4]) * dr;
                inv[7] = (It[6] * It[1] - It[0] * It[7]) * dr;
                inv[8] = (It[0] * It[4] - It[1] * It[3]) * dr;
                float M[9], tau[3];
                mul0_333(M, I, inv);
                M[0] -= 1; M[4] -= 1; M[8] -= 1;
                mul0_331(tau, M, L);
                b.tacc[0] += tau[0]; b.tacc[1] += tau[1]; b.tacc[2] += tau[2];
            }
        }
        b.facc[0] += b.mass * gravity[0];
        b.facc[1] += b.mass * gravity[1];
        b.facc[2] += b.mass * gravity[2];
    }

    // rows
    int m = 0;
    for (int j : jointOrder) m += joints[j].rows();
    lastM = m;
    static thread_local std::vector<Row> rows; rows.assign(m > 0 ? m : 1, Row());
    static thread_local std::vector<int> rb0, rb1, jofs; rb0.assign(m, 0); rb1.assign(m, 0); jofs.assign(jointOrder.size() + 1, 0);
    {
        int o = 0;
        for (size_t t = 0; t < jointOrder.size(); ++t) {
            const Joint& j = joints[jointOrder[t]];
            jofs[t] = o;
            const int mm = jointRows(*this, j, fps, &rows[o]);
            for (int r = 0; r < mm; ++r) { rb0[o + r] = j.b0; rb1[o + r] = j.b1; }
            o += mm;
        }
        jofs[jointOrder.size()] = o;
    }

    // tmp1 = v*fps + invM fe
    static thread_local std::vector<float> tmp1; tmp1.assign(nb * 6, float());
    for (int bi = 0; bi < nb; ++bi) {
        const Body& b = bodies[bi];
        float* t = &tmp1[bi * 6];
        for (int k = 0; k < 3; ++k) t[k] = b.facc[k] * b.invMass + b.lvel[k] * fps;
        mul0_331(t + 3, &invIw[bi * 9], b.tacc);
        for (int k = 0; k < 3; ++k) t[3 + k] += b.avel[k] * fps;
    }
    static thread_local std::vector<float> lambda; lambda.assign(m, 0.0f);
    static thread_local std::vector<float> JinvM, Am, dinv;
    if (m > 0) {
        // JinvM
        JinvM.assign(m * 12, float());
        for (int i = 0; i < m; ++i) {
            const float* J = rows[i].J;
            float* o = &JinvM[i * 12];
            const Body& A = bodies[rb0[i]];
            const Body& B = bodies[rb1[i]];
            const float* ia = &invIw[rb0[i] * 9];
            const float* ib = &invIw[rb1[i] * 9];
            for (int k = 0; k < 3; ++k) { o[k] = J[k] * A.invMass; o[6 + k] = J[6 + k] * B.invMass; }
            // row-vector times matrix (dMultiply0_133)
            for (int c2 = 0; c2 < 3; ++c2) {
                o[3 + c2] = J[3] * ia[0 * 3 + c2] + J[4] * ia[1 * 3 + c2] + J[5] * ia[2 * 3 + c2];
                o[9 + c2] = J[9] * ib[0 * 3 + c2] + J[10] * ib[1 * 3 + c2] + J[11] * ib[2 * 3 + c2];
            }
        }
        // A = JinvM J^T (lower triangle), + cfm*fps on the diagonal
        Am.assign(m * m, 0.0f);
        auto dot6 = [](const float* a, const float* b) {
            return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
        };
        for (int i = 0; i < m; ++i) {
            for (int jx = 0; jx <= i; ++jx) {
                float s = 0.0f;
                bool any = false;
                const int ib[2] = {rb0[i], rb1[i]};
                const int jb[2] = {rb0[jx], rb1[jx]};
                for (int si = 0; si < 2; ++si)
                    for (int sj = 0; sj < 2; ++sj)
                        if (ib[si] == jb[sj]) {
                            const float d = dot6(&JinvM[i * 12 + si * 6], &rows[jx].J[sj * 6]);
                            s = any ? (s + d) : d;
                            any = true;
                        }
                Am[i * m + jx] = s;
            }
            Am[i * m + i] += rows[i].cfm * fps;
        }
        // rhs = c*fps - J tmp1
        static thread_local std::vector<float> rhs; rhs.assign(m, float());
        for (int i = 0; i < m; ++i) {
            float r = rows[i].c * fps;
            r -= dot6(&rows[i].J[0], &tmp1[rb0[i] * 6]);
            r -= dot6(&rows[i].J[6], &tmp1[rb1[i] * 6]);
            rhs[i] = r;
        }
        lastA = Am;
        lastRhs = rhs;
        // right-looking LDL^T, lower triangle, explicit fmaf
        dinv.assign(m, float());
        for (int k = 0; k < m; ++k) {
            const float d = Am[k * m + k];
            const float id = 1.0f / d;
            dinv[k] = id;
            for (int i = k + 1; i < m; ++i) {
                const float lik = Am[i * m + k] * id;
                for (int jx = k + 1; jx <= i; ++jx)
                    Am[i * m + jx] = fmaf(-lik, Am[jx * m + k], Am[i * m + jx]);
            }
            for (int i = k + 1; i < m; ++i) Am[i * m + k] *= id;  // store L
        }
        // forward (column oriented), diagonal, backward (column oriented)
        for (int k = 0; k < m; ++k)
            for (int i = k + 1; i < m; ++i) rhs[i] = fmaf(-Am[i * m + k], rhs[k], rhs[i]);
        for (int k = 0; k < m; ++k) rhs[k] *= dinv[k];
        for (int k = m - 1; k >= 0; --k)
            for (int i = 0; i < k; ++i) rhs[i] = fmaf(-Am[k * m + i], rhs[k], rhs[i]);
        lambda = rhs;
    }
    float contactForce[6] = {0, 0, 0, 0, 0, 0};
    lastContactLambda.clear(); lastContactLo.clear(); lastContactHi.clear(); lastLcpIterations = 0;
    if (!contacts.empty()) solveContacts(*this, fps, m, rb0.data(), rb1.data(), JinvM.data(), Am.data(), dinv.data(), &invIw[contactBody * 9], &tmp1[contactBody * 6], lambda.data(), contactForce);
    lastLambda = lambda;

    // cforce = J^T lambda (per joint, per component: sum over the joint's rows, then accumulate)
    static thread_local std::vector<float> cf; cf.assign(nb * 6, 0.0f);
    for (size_t t = 0; t < jointOrder.size(); ++t) {
        const Joint& j = joints[jointOrder[t]];
        const int o = jofs[t], mm = jofs[t + 1] - jofs[t];
        for (int k = 0; k < 6; ++k) {
            float s0 = 0.0f, s1 = 0.0f;
            for (int r = 0; r < mm; ++r) {
                s0 += rows[o + r].J[k] * lambda[o + r];
                s1 += rows[o + r].J[6 + k] * lambda[o + r];
            }
            cf[j.b0 * 6 + k] += s0;
            cf[j.b1 * 6 + k] += s1;
        }
    }
    for (int k = 0; k < 6; ++k) cf[contactBody * 6 + k] += contactForce[k];   // the contact joints come last (their group is the newest)
    // velocity update, position update (dxStepBody, finite rotation mode 1, no finite-rotation axis:
    // RigidBodyODE.cpp:15-16), zero accumulators
    for (int bi = 0; bi < nb; ++bi) {
        Body& b = bodies[bi];
        const float* c = &cf[bi * 6];
        const float ims = h * b.invMass;
        for (int k = 0; k < 3; ++k) b.lvel[k] += (c[k] + b.facc[k]) * ims;
        float tt[3], dw[3];
        for (int k = 0; k < 3; ++k) tt[k] = (c[3 + k] + b.tacc[k]) * h;
        mul0_331(dw, &invIw[bi * 9], tt);
        for (int k = 0; k < 3; ++k) b.avel[k] += dw[k];

        for (int k = 0; k < 3; ++k) b.pos[k] += h * b.lvel[k];
        {
            const float wlen = sqrtf(b.avel[0] * b.avel[0] + b.avel[1] * b.avel[1] + b.avel[2] * b.avel[2]);
            const float hh = h * 0.5f;
            const float theta = wlen * hh;
            float qr[4], q2[4];
            qr[0] = m_cosf(theta);
            const float s = sinc_ode(theta) * hh;
            qr[1] = b.avel[0] * s; qr[2] = b.avel[1] * s; qr[3] = b.avel[2] * s;
            qmul0(q2, qr, b.q);
            b.q[0] = q2[0]; b.q[1] = q2[1]; b.q[2] = q2[2]; b.q[3] = q2[3];
        }
        normalize4(b.q);
        rFromQ(b.R, b.q);
        for (int k = 0; k < 3; ++k) { b.facc[k] = 0; b.tacc[k] = 0; }
    }
}

}  // namespace pdrb
