// ORACLE / TEST INFRASTRUCTURE.
#pragma once
#include "Physics/IPhysicsEngine.h"
#include "../rb/pdrb.h"
namespace D {
pdrb::World* ref_get_world(IPhysicsEngine* p);
int ref_body_id(IRigidBody* b);
void ref_set_collide(IPhysicsEngine* p, bool on);   // run the engine's collision pass (contacts from oracle/rb/pdcollide.h)
void ref_enable_log(bool v);
}
