// ORACLE / TEST INFRASTRUCTURE (fixture generator; runs only in the build container).
//
// Implementation of the reference's own plugin seam Physics/IPhysicsEngine.h:10-28 (selected by
// PhysicsFactory::createPhysicsEngine, Physics/PhysicsFactory.cpp:6-9) on top of this project's
// rigid-body restatement oracle/rb (pdrb).  It plays the role of Physics/ODE/*.cpp, which cannot
// be built here because libode is absent.  Method-by-method it follows RigidBodyODE.cpp:9-319,
// JointODE.cpp:21-89, PhysicsEngineODE.cpp:111-341.  Body-vs-track collision (collisionStep / collisionNearCallback /
// onCollision, PhysicsEngineODE.cpp:228-341) is optional (ref_set_collide): the frame parity, the category / mask pairing, the
// body-local normal.y filter of box contacts and the callback into the reference's Simulator / Car::onCollisionCallback follow
// the reference; the contacts themselves come from this project's own generator (oracle/rb/pdcollide.h -- ODE's dCollide is
// not available).  onCollision's contact joints (:283-331) go into pdrb::World::contacts (the body's deepest
// pdcol::MAX_CONTACTS contact points, alive until the next odd frame refills the group); pdrb solves them.
#include "Physics/PhysicsFactory.h"
#include "Physics/IPhysicsEngine.h"
#include "Core/Diag.h"
#include "../rb/pdrb.h"
#include "../rb/pdray.h"
#include "../rb/pdcollide.h"
#include <cfloat>
#include "ref_physics.h"
#include <vector>
#include <algorithm>
#include <cstring>
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
    void addBoxCollider(const vec3f& pos, const vec3f& size, unsigned int, unsigned int category, unsigned long mask) override;
    void addMeshCollider(ITriMeshPtr tm, const mat44f& offset, unsigned int, unsigned long category, unsigned long mask) override;
};
// geoms attached to a body (RigidBodyODE::addBoxCollider / addMeshCollider, RigidBodyODE.cpp:274-317), in the body frame
struct DynBox { BodyPD* body; float centre[3], half[3]; unsigned long cat, mask; };
struct DynMesh { BodyPD* body; std::vector<float> verts; std::vector<unsigned char> tris; std::shared_ptr<ColliderPD> shape; };

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
    std::vector<DynBox> boxes;
    std::vector<DynMesh> meshes;
    bool collide = false;
    int currentFrame = 0;
    pdcol::ContactSet contactSet;   // contactGroupDynamic (one car per simulator: one body carries every contact)
    void collisionStep();

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
    void step(float dt) override { if (collide) collisionStep(); else currentFrame++; world.step(dt); }  // PhysicsEngineODE.cpp:216-224
};

pdrb::Body& BodyPD::b() { return e->world.bodies[id]; }
void BodyPD::addBoxCollider(const vec3f& pos, const vec3f& size, unsigned int, unsigned int category, unsigned long mask) {
    DynBox bx; bx.body = this; bx.cat = category; bx.mask = mask;
    bx.centre[0] = pos.x; bx.centre[1] = pos.y; bx.centre[2] = pos.z;
    bx.half[0] = size.x * 0.5f; bx.half[1] = size.y * 0.5f; bx.half[2] = size.z * 0.5f;
    e->boxes.push_back(bx);
}
void BodyPD::addMeshCollider(ITriMeshPtr tm, const mat44f& o, unsigned int, unsigned long category, unsigned long mask) {
    // geom offset: rotation = transposed 3x3 of `offset`, position = its fourth row (RigidBodyODE.cpp:300-314): a mesh
    // vertex v sits at v * offset (row vector) in the body frame
    auto t = std::dynamic_pointer_cast<TriMeshPD>(tm);
    DynMesh m; m.body = this;
    m.shape = std::make_shared<ColliderPD>(); m.shape->cat = category; m.shape->mask = mask;
    const float M[9] = {o.M11, o.M12, o.M13, o.M21, o.M22, o.M23, o.M31, o.M32, o.M33};
    const float off[3] = {o.M41, o.M42, o.M43};
    for (size_t i = 0; i < t->vb.size(); ++i) {
        const float v[3] = {t->vb[i].x, t->vb[i].y, t->vb[i].z};
        for (int c = 0; c < 3; ++c) m.verts.push_back(off[c] + (v[0] * M[c] + v[1] * M[3 + c] + v[2] * M[6 + c]));
    }
    for (size_t i = 0; i < t->ib.size(); ++i) m.tris.push_back((unsigned char)t->ib[i]);
    if (t->vb.size() > 256) SHOULD_NOT_REACH_FATAL;
    e->meshes.push_back(std::move(m));
}
void EnginePD::collisionStep() {
    const int frame = currentFrame++;
    if (!(frame & 1)) return;   // even frames collide dynamic geoms with each other (one car: nothing), odd ones dynamic vs static
    contactSet.clear();         // dJointGroupEmpty(contactGroupDynamic)
    world.contacts.clear();
    // one broad-phase box per body: around all of its geoms, in the body frame
    std::vector<BodyPD*> bodies;
    for (auto& b : boxes) if (std::find(bodies.begin(), bodies.end(), b.body) == bodies.end()) bodies.push_back(b.body);
    for (auto& m : meshes) if (std::find(bodies.begin(), bodies.end(), m.body) == bodies.end()) bodies.push_back(m.body);
    for (BodyPD* body : bodies) {
        float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        auto grow = [&](const float* v) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], v[k]); hi[k] = std::max(hi[k], v[k]); } };
        for (auto& b : boxes) if (b.body == body)
            for (int c = 0; c < 8; ++c) {
                const float v[3] = {b.centre[0] + ((c & 1) ? b.half[0] : -b.half[0]), b.centre[1] + ((c & 2) ? b.half[1] : -b.half[1]), b.centre[2] + ((c & 4) ? b.half[2] : -b.half[2])};
                grow(v);
            }
        for (auto& m : meshes) if (m.body == body) for (size_t i = 0; i + 2 < m.verts.size(); i += 3) grow(&m.verts[i]);
        pdcol::Pose pose;
        memcpy(pose.pos, body->b().pos, 12); memcpy(pose.R, body->b().R, 36);
        pdcol::V aLo, aHi;
        pdcol::worldAabb(pose, lo, hi, aLo, aHi);
        unsigned triBase = 0;   // triangle ids run over the static meshes in creation order (= the track blob's triangle order)
        for (size_t si = 0; si < statics.size(); triBase += (unsigned)(statics[si].indices.size() / 3), ++si) {
            const pdrb::StaticMesh& sm = statics[si];
            for (size_t ti = 0; ti + 2 < sm.indices.size(); ti += 3) {
                const unsigned tid = (triBase + (unsigned)(ti / 3)) * pdcol::ID_STRIDE;
                const pdcol::V p0 = pdcol::ld(&sm.verts[3 * sm.indices[ti]]), p1 = pdcol::ld(&sm.verts[3 * sm.indices[ti + 1]]), p2 = pdcol::ld(&sm.verts[3 * sm.indices[ti + 2]]);
                if (!pdcol::triMeetsAabb(p0, p1, p2, aLo, aHi)) continue;
                unsigned boxIndex = 0;   // which of the body's box geoms, in creation order: sixteen contact-item ids per box
                for (auto& b : boxes) {
                    if (b.body != body) continue;
                    const unsigned bx = boxIndex++;
                    if (!((b.cat & sm.mask) && (sm.category & b.mask))) continue;   // collisionNearCallback, :258-264
                    float ny;
                    if (!pdcol::boxContacts(pose, b.centre, b.half, p0, p1, p2, ny, [&](const pdcol::V& pw, const pdcol::V& nw, float depth, int item) {
                            contactSet.insert(pw, nw, depth, 1, tid + (unsigned)item + 16u * bx);
                        }) || ny < 0.9f) continue;   // onCollision, :303-312
                    pdcol::V n = pdcol::norm(pdcol::cross(p1 - p0, p2 - p0));
                    const pdcol::V cw = pdcol::toWorld(pose, pdcol::ld(b.centre));
                    if (pdcol::dot(n, cw - p0) < 0.0f) n = n * -1.0f;
                    if (cb) cb->onCollisionCallback(body, nullptr, nullptr, staticColliders[si].get(), vec3f(n.x, n.y, n.z), vec3f(cw.x, cw.y, cw.z), 0.0f);   // a box geom carries no shape data
                }
                bool boundsKnown = false, inBounds = false;
                for (auto& m : meshes) {
                    if (m.body != body || !((m.shape->cat & sm.mask) && (sm.category & m.shape->mask))) continue;
                    if (!boundsKnown) { inBounds = pdcol::triMeetsBounds(pose, lo, hi, p0, p1, p2); boundsKnown = true; }   // pdcollide.h: a triangle outside the colliders' box meets no hull
                    if (!inBounds) continue;
                    pdcol::hullContacts(pose, reinterpret_cast<const float (*)[3]>(m.verts.data()), reinterpret_cast<const unsigned char (*)[3]>(m.tris.data()), (int)(m.tris.size() / 3),
                                        p0, p1, p2, [&](const pdcol::V& n, const pdcol::V& hit, float depth, int item) {
                        contactSet.insert(hit, n, depth, 0, tid + (unsigned)item);
                        if (cb) cb->onCollisionCallback(body, m.shape.get(), nullptr, staticColliders[si].get(), vec3f(n.x, n.y, n.z), vec3f(hit.x, hit.y, hit.z), 0.0f);
                    });
                }
            }
        }
        world.contactBody = body->id;
    }
    world.contacts.resize(contactSet.n);
    for (int i = 0; i < contactSet.n; ++i) memcpy(&world.contacts[i], &contactSet.c[i], sizeof(pdrb::ContactJoint));
}

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
void ref_set_collide(IPhysicsEngine* p, bool on) { static_cast<EnginePD*>(p)->collide = on; }
int ref_body_id(IRigidBody* b) { return static_cast<BodyPD*>(b)->id; }

}  // namespace D
