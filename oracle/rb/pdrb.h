// ORACLE / TEST INFRASTRUCTURE -- not product code.  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may link or run anything under oracle/.
//
// pdrb: scalar single-precision restatement of the rigid-body arithmetic the reference
// obtains from Open Dynamics Engine 0.16.3 (single precision, dWorldStep = exact
// "big-matrix" stepper).  ODE itself is NOT in /root/reference (only headers under
// thirdparty/ode/include; the library is a missing blob), so this file restates ODE's
// published algorithm anchored on the reference's call sites:
//   Physics/ODE/PhysicsEngineODE.cpp:23-29,216-224  (world parameters, dWorldStep)
//   Physics/ODE/RigidBodyODE.cpp:9-270              (body API used by the car code)
//   Physics/ODE/JointODE.cpp:21-89                  (Fixed / Ball / Slider / DBall joints)
// and on the inline arithmetic that DOES ship in thirdparty/ode/include/ode/odemath.h
// (dot/cross evaluation order :213-243, cross-matrix signs :277-297, dInvertMatrix3 :463-500).
//
// PARITY UNPINNED for this file: the reference holds no test, fixture or golden vector at the
// ODE boundary and the library cannot be run here.  Row ordering, LDLT elimination order and
// normalisation details are this project's canonical choices (documented in DESIGN.md).
#pragma once
#include <vector>
#include <cstdint>
#include <cstddef>

namespace pdrb {

struct Body {
    float pos[3] = {0, 0, 0};
    float q[4] = {1, 0, 0, 0};
    float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // row-major 3x3 (ODE dMatrix3 without the pad column)
    float lvel[3] = {0, 0, 0};
    float avel[3] = {0, 0, 0};
    float facc[3] = {0, 0, 0};
    float tacc[3] = {0, 0, 0};
    float mass = 1.0f;
    float invMass = 1.0f;
    float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};     // body-frame inertia
    float invI[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // body-frame inverse inertia

    // --- mass (RigidBodyODE.cpp:64-98) ---
    void setMassBoxTotal(float m, float lx, float ly, float lz);
    // --- pose ---
    void setPosition(float x, float y, float z) { pos[0] = x; pos[1] = y; pos[2] = z; }
    void setRotation(const float Rin[9]);  // dBodySetRotation: orthogonalise, q from R
    // --- frame transforms (RigidBodyODE.cpp:101-127) ---
    void relPointPos(const float p[3], float out[3]) const;    // dBodyGetRelPointPos
    void posRelPoint(const float p[3], float out[3]) const;    // dBodyGetPosRelPoint
    void vectorToWorld(const float v[3], float out[3]) const;  // dBodyVectorToWorld
    void vectorFromWorld(const float v[3], float out[3]) const;
    void relPointVel(const float p[3], float out[3]) const;    // dBodyGetRelPointVel (p body-local)
    void pointVel(const float p[3], float out[3]) const;       // dBodyGetPointVel (p world)
    // --- accumulators (RigidBodyODE.cpp:229-268) ---
    void addForceAtPos(const float f[3], const float p[3]);         // f world, p world
    void addForceAtRelPos(const float f[3], const float p[3]);      // f world, p local
    void addRelForceAtRelPos(const float f[3], const float p[3]);   // f local, p local
    void addRelForceAtPos(const float f[3], const float p[3]);      // f local, p world
    void addTorque(const float t[3]);
    void addRelTorque(const float t[3]);
    void stop();  // RigidBodyODE.cpp:55-61
};

enum JointType { JT_FIXED = 0, JT_BALL = 1, JT_SLIDER = 2, JT_DBALL = 3 };

struct Joint {
    int type = JT_BALL;
    int b0 = -1, b1 = -1;
    float erp = 0.3f, cfm = 1e-7f;   // joint-level ERP/CFM (dball; ball/fixed copy the world's at creation)
    float anchor1[3] = {0, 0, 0};    // body0-local
    float anchor2[3] = {0, 0, 0};    // body1-local
    float axis1[3] = {1, 0, 0};      // slider axis, body0-local
    float offset[3] = {0, 0, 0};     // fixed: R0^T (p0-p1); slider: R1^T (p0-p1)
    float qrel[4] = {1, 0, 0, 0};    // inv(q0) * q1 at attach time
    float targetDistance = 0;        // dball
    int rows() const { return type == JT_FIXED ? 6 : type == JT_BALL ? 3 : type == JT_SLIDER ? 5 : 1; }
};

// A contact joint of the current contact group (PhysicsEngineODE::onCollision, PhysicsEngineODE.cpp:283-331): always between
// body `World::contactBody` and the static world.  kind 0 = every pair but box-trimesh (mode 28692: Bounce | SoftCFM | Approx1,
// mu 0.25, bounce 0.01, soft_cfm 1e-4); kind 1 = box-trimesh (mode 28700: + SoftERP, mu 0.1, bounce 0, soft_cfm 0.000952380942,
// soft_erp 0.714285731).
struct ContactJoint { float pos[3]; float depth; float normal[3]; int kind; };

struct World {
    float gravity[3] = {0.0f, -9.80665f, 0.0f};  // PhysicsEngineODE.cpp:23
    float erp = 0.3f;                            // :24
    float cfm = 1.0e-7f;                         // :25
    float contactMaxCorrectingVel = 3.0f;        // :26
    float contactSurfaceLayer = 0.0f;            // :27
    std::vector<ContactJoint> contacts;          // contactGroupDynamic: refilled on odd frames, alive until then (:228-243)
    int contactBody = 0;
    std::vector<Body> bodies;
    std::vector<Joint> joints;
    std::vector<int> jointOrder;  // island traversal order (see buildOrder)
    bool orderDirty = true;

    int createBody() { bodies.emplace_back(); orderDirty = true; return (int)bodies.size() - 1; }
    // JointODE.cpp:21-60 (positions in WORLD coordinates, like the ODE setters)
    int createFixed(int b0, int b1);
    int createBall(int b0, int b1, const float anchorWorld[3]);
    int createSlider(int b0, int b1, const float axisWorld[3]);
    int createDBall(int b0, int b1, const float a1World[3], const float a2World[3]);
    // JointODE.cpp:62-89
    void dballSetAnchor1(int j, const float w[3]);
    void dballSetAnchor2(int j, const float w[3]);
    void dballUpdateTargetDistance(int j);

    void buildOrder();
    void step(float h);  // dWorldStep

    // scratch exposed for tests (filled by step)
    int lastM = 0;
    std::vector<float> lastLambda;
    std::vector<float> lastA;  // m*m, before factorisation (lower triangle meaningful)
    std::vector<float> lastRhs;
    // contact rows of the last step (3 per contact: normal, friction 1, friction 2), for the invariants tests
    std::vector<float> lastContactLambda, lastContactLo, lastContactHi;
    int lastLcpIterations = 0;
};

// shared small math (also used by the oracle's car code so evaluation order is identical)
inline float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline void cross3(float* r, const float* a, const float* b) {
    const float r0 = a[1] * b[2] - a[2] * b[1];
    const float r1 = a[2] * b[0] - a[0] * b[2];
    const float r2 = a[0] * b[1] - a[1] * b[0];
    r[0] = r0; r[1] = r1; r[2] = r2;
}
void mul0_331(float* r, const float* M, const float* v);  // r = M v
void mul1_331(float* r, const float* M, const float* v);  // r = M^T v
void qFromR(float q[4], const float R[9]);
void rFromQ(float R[9], const float q[4]);
void normalize4(float q[4]);
void normalize3(float v[3]);
void planeSpace(const float n[3], float p[3], float q[3]);

}  // namespace pdrb
