// ORACLE / TEST INFRASTRUCTURE (fixture generator; runs only in the build container).
//
// Implementation of the reference's own plugin seam Physics/IPhysicsEngine.h:10-28 (selected by
// PhysicsFactory::createPhysicsEngine, Physics/PhysicsFactory.cpp:6-9) on top of this project's
// rigid-body restatement oracle/rb (pdrb).  It plays the role of Physics/ODE/*.cpp, which cannot
// be built here because libode is absent.  Method-by-method it follows RigidBodyODE.cpp:9-319,
// JointODE.cpp:21-89, PhysicsEngineODE.cpp:111-224.  Body-vs-track collision (collisionStep,
// PhysicsEngineODE.cpp:228-341) is NOT implemented: colliders are recorded and ignored.
#include "Physics/PhysicsFactory.h"
#include "Physics/IPhysicsEngine.h"
#include "Core/Diag.h"
#include "../rb/pdrb.h"
#include "../rb/pdray.h"
#include "ref_physics.h"
#include <vector>
#include <memory>

namespace D {

struct TriMeshPD : public ITriMesh {
    std::vector<TriMeshVertex> vb;
    std::vector<TriMeshIndex> ib;
    void resize(size_t v, size_t i) override { vb.resize(v); ib.resize(i); }
    TriMeshVertex* getVB() override { return vb.data(); }
    size_t getVertexCount() override { return vb.size(); }
    TriMeshIndex* getIB() override { return ib.data(); }
    size_t getIndexCount() override { return ib.size(); }
};

struct ColliderPD : public ICollisionObject {
    void* user = nullptr;
    unsigned long cat = 0, mask = 0;
    void setUserPointer(void* d) override { user = d; }
    void* getUserPointer() override { return user; }
    unsigned long getGroup() override { return cat; }
    unsigned long getMask() override { return mask; }
};

struct EnginePD;

struct BodyPD : public IRigidBody {
    EnginePD* e;
    int id;
    BodyPD(EnginePD* _e, int _id) : e(_e), id(_id) {}
    pdrb::Body& b();
    void setEnabled(bool) override {}
    bool isEnabled() override { return true; }
    void setAutoDisable(bool) override {}
    void stop() override { b().stop(); }
    void setMassBox(float m, float x, float y, float z) override { b().setMassBoxTotal(m, x, y, z); }
    float getMass() override { return b().mass; }
    void setMassExplicitInertia(float, float, float, float) override { SHOULD_NOT_REACH_FATAL; }
    vec3f getLocalInertia() override { return vec3f(b().I[0], b().I[4], b().I[8]); }
    vec3f localToWorld(const vec3f& p) override { float o[3]; b().relPointPos(&p.x, o); return vec3f(o); }
    vec3f worldToLocal(const vec3f& p) override { float o[3]; b().posRelPoint(&p.x, o); return vec3f(o); }
    vec3f localToWorldNormal(const vec3f& p) override { float o[3]; b().vectorToWorld(&p.x, o); return vec3f(o); }
    vec3f worldToLocalNormal(const vec3f& p) override { float o[3]; b().vectorFromWorld(&p.x, o); return vec3f(o); }
    void setPosition(const vec3f& p) override { b().setPosition(p.x, p.y, p.z); }
    vec3f getPosition(float) override { return vec3f(b().pos); }
    void setRotation(const mat44f& m) override {
        // RigidBodyODE.cpp:140-156
        const float R[9] = {m.M11, m.M21, m.M31, m.M12, m.M22, m.M32, m.M13, m.M23, m.M33};
        b().setRotation(R);
    }
    mat44f getWorldMatrix(float) override {
        // RigidBodyODE.cpp:158-180
        const float* r = b().R;
        const float* p = b().pos;
        mat44f m;
        m.M11 = r[0]; m.M12 = r[3]; m.M13 = r[6]; m.M14 = 0;
        m.M21 = r[1]; m.M22 = r[4]; m.M23 = r[7]; m.M24 = 0;
        m.M31 = r[2]; m.M32 = r[5]; m.M33 = r[8]; m.M34 = 0;
        m.M41 = p[0]; m.M42 = p[1]; m.M43 = p[2]; m.M44 = 1.0f;
        return m;
    }
    void setVelocity(const vec3f& v) override { b().lvel[0] = v.x; b().lvel[1] = v.y; b().lvel[2] = v.z; }
    vec3f getVelocity() override { const float z[3] = {0, 0, 0}; float o[3]; b().relPointVel(z, o); return vec3f(o); }
    vec3f getLocalVelocity() override { return worldToLocalNormal(getVelocity()); }
    vec3f getPointVelocity(const vec3f& p) override { float o[3]; b().pointVel(&p.x, o); return vec3f(o); }
    vec3f getLocalPointVelocity(const vec3f& p) override { float o[3]; b().relPointVel(&p.x, o); return vec3f(o); }
    void setAngularVelocity(const vec3f& v) override { b().avel[0] = v.x; b().avel[1] = v.y; b().avel[2] = v.z; }
    vec3f getAngularVelocity() override { return vec3f(b().avel); }
    vec3f getLocalAngularVelocity() override { return worldToLocalNormal(getAngularVelocity()); }
    void addForceAtPos(const vec3f& f, const vec3f& p) override { b().addForceAtPos(&f.x, &p.x); }
    void addForceAtLocalPos(const vec3f& f, const vec3f& p) override { b().addForceAtRelPos(&f.x, &p.x); }
    void addLocalForce(const vec3f& f) override { const float z[3] = {0, 0, 0}; b().addRelForceAtRelPos(&f.x, z); }
    void addLocalForceAtPos(const vec3f& f, const vec3f& p) override { b().addRelForceAtPos(&f.x, &p.x); }
    void addLocalForceAtLocalPos(const vec3f& f, const vec3f& p) override { b().addRelForceAtRelPos(&f.x, &p.x); }
    void addTorque(const vec3f& t) override { b().addTorque(&t.x); }
    void addLocalTorque(const vec3f& t) override { b().addRelTorque(&t.x); }
    void addBoxCollider(const vec3f&, const vec3f&, unsigned int, unsigned int, unsigned long) override {}
    void addMeshCollider(ITriMeshPtr, const mat44f&, unsigned int, unsigned long, unsigned long) override {}
};

struct JointPD : public IJoint {
    EnginePD* e;
    int id;
    int kind;  // pdrb::JointType
    float distance = 0;
    JointPD(EnginePD* _e, int _id, int k) : e(_e), id(_id), kind(k) {}
    void setERPCFM(float erp, float cfm) override;
    void reseatDistanceJointLocal(const vec3f& p1, const vec3f& p2) override;
};

struct RayCasterPD : public IRayCaster {
    EnginePD* e;
    float length;
    RayCasterPD(EnginePD* _e, float l) : e(_e), length(l) {}
    RayCastHit rayCast(const vec3f& pos, const vec3f& dir) override;
};

struct EnginePD : public IPhysicsEngine {
    pdrb::World world;
    std::vector<pdrb::StaticMesh> statics;
    std::vector<std::shared_ptr<ColliderPD>> staticColliders;
    ICollisionCallback* cb = nullptr;

    IRigidBodyPtr createRigidBody() override { return std::make_shared<BodyPD>(this, world.createBody()); }
    static int bid(const IRigidBodyPtr& p) { return std::dynamic_pointer_cast<BodyPD>(p)->id; }
    IJointPtr createFixedJoint(IRigidBodyPtr a, IRigidBodyPtr b) override {
        return std::make_shared<JointPD>(this, world.createFixed(bid(a), bid(b)), pdrb::JT_FIXED);
    }
    IJointPtr createBallJoint(IRigidBodyPtr a, IRigidBodyPtr b, const vec3f& pos) override {
        return std::make_shared<JointPD>(this, world.createBall(bid(a), bid(b), &pos.x), pdrb::JT_BALL);
    }
    IJointPtr createSliderJoint(IRigidBodyPtr a, IRigidBodyPtr b, const vec3f& axis) override {
        return std::make_shared<JointPD>(this, world.createSlider(bid(a), bid(b), &axis.x), pdrb::JT_SLIDER);
    }
    IJointPtr createDistanceJoint(IRigidBodyPtr a, IRigidBodyPtr b, const vec3f& p1, const vec3f& p2) override {
        auto j = std::make_shared<JointPD>(this, world.createDBall(bid(a), bid(b), &p1.x, &p2.x), pdrb::JT_DBALL);
        j->distance = world.joints[j->id].targetDistance;  // JointODE.cpp:59
        return j;
    }
    IJointPtr createBumpJoint(IRigidBodyPtr, IRigidBodyPtr, const vec3f&, float, float) override { return nullptr; }
    ITriMeshPtr createTriMesh() override { return std::make_shared<TriMeshPD>(); }
    ICollisionObjectPtr createCollider(ITriMeshPtr tm, bool isDynamic, unsigned int, unsigned long category, unsigned long mask) override {
        auto c = std::make_shared<ColliderPD>();
        c->cat = category; c->mask = mask;
        if (!isDynamic) {
            auto t = std::dynamic_pointer_cast<TriMeshPD>(tm);
            pdrb::StaticMesh sm;
            sm.verts.resize(t->vb.size() * 3);
            for (size_t i = 0; i < t->vb.size(); ++i) { sm.verts[3 * i] = t->vb[i].x; sm.verts[3 * i + 1] = t->vb[i].y; sm.verts[3 * i + 2] = t->vb[i].z; }
            sm.indices.assign(t->ib.begin(), t->ib.end());
            sm.category = category; sm.mask = mask;
            statics.push_back(std::move(sm));
            staticColliders.push_back(c);
        }
        return c;
    }
    IRayCasterPtr createRayCaster(float length) override { return std::make_shared<RayCasterPD>(this, length); }
    void setCollisionCallback(ICollisionCallback* c) override { cb = c; }
    RayCastHit rayImpl(const vec3f& pos, const vec3f& dir, float length) {
        RayCastHit hit;
        pdrb::RayHit h = pdrb::rayCastMeshes(statics, &pos.x, &dir.x, length);
        if (h.has) {
            hit.pos = vec3f(h.pos);
            hit.normal = vec3f(h.normal);
            hit.collisionObject = staticColliders[h.mesh].get();
            hit.hasContact = true;
        }
        return hit;
    }
    RayCastHit rayCast(const vec3f& pos, const vec3f& dir, float length) override { return rayImpl(pos, dir, length); }
    RayCastHit rayCast(const vec3f& pos, const vec3f& dir, IRayCasterPtr ray) override {
        return rayImpl(pos, dir, std::dynamic_pointer_cast<RayCasterPD>(ray)->length);
    }
    void step(float dt) override { world.step(dt); }  // PhysicsEngineODE.cpp:216-224 without collisionStep
};

pdrb::Body& BodyPD::b() { return e->world.bodies[id]; }

void JointPD::setERPCFM(float erp, float cfm) {
    // Only SliderJointODE / DistanceJointODE override setERPCFM (JointODE.h:17-44).  The slider's
    // dParamERP is not a limit-motor parameter and dParamCFM only reaches the (absent) motor row.
    if (kind == pdrb::JT_DBALL) {
        if (erp > 0.0f) e->world.joints[id].erp = erp;
        if (cfm > 0.0f) e->world.joints[id].cfm = cfm;
    }
}
void JointPD::reseatDistanceJointLocal(const vec3f& p1, const vec3f& p2) {
    if (kind != pdrb::JT_DBALL) return;
    // JointODE.cpp:77-89
    pdrb::Joint& j = e->world.joints[id];
    float w1[3], w2[3];
    e->world.bodies[j.b0].relPointPos(&p1.x, w1);
    e->world.bodies[j.b1].relPointPos(&p2.x, w2);
    e->world.dballSetAnchor1(id, w1);
    e->world.dballSetAnchor2(id, w2);
    j.targetDistance = distance;
}
RayCastHit RayCasterPD::rayCast(const vec3f& pos, const vec3f& dir) { return e->rayImpl(pos, dir, length); }

std::shared_ptr<IPhysicsEngine> PhysicsFactory::createPhysicsEngine() { return std::make_shared<EnginePD>(); }

pdrb::World* ref_get_world(IPhysicsEngine* p) { return &static_cast<EnginePD*>(p)->world; }
int ref_body_id(IRigidBody* b) { return static_cast<BodyPD*>(b)->id; }

}  // namespace D
