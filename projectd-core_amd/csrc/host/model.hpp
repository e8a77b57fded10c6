// pdbatch host side: model / track builders and reset-pose helpers (see model.cpp, track.cpp, reset.cpp).
#pragma once
#include <string>
#include <vector>
#include <stdexcept>
#include "pdb_types.h"

namespace pdb {

void buildCarModel(const std::string& basePath, const std::string& modelName, pdb_car_params& out);
bool setCarTune(pdb_car_params& P, const std::string& basePath, const std::string& modelName, const std::string& name, float value, bool raw);
bool setScoringVar(pdb_car_params& P, const std::string& name, float w);
bool getScoringVar(const pdb_car_params& P, const std::string& name, float& w);

// track blob (pdb_track_header + arrays)
std::vector<uint8_t> buildTrack(const std::string& basePath, const std::string& trackName, bool recomputeFatPoints = false);

struct TrackView {
    const pdb_track_header* h = nullptr;
    const pdb_surface* surfaces = nullptr;
    const float* tris = nullptr;
    const float* fat = nullptr;
    const float* fatDist = nullptr;
    const float* nodes = nullptr;
    const float* nodeDist = nullptr;
    const float* pits = nullptr;   // float[h->numPits][16]
    explicit TrackView(const uint8_t* blob) {
        h = reinterpret_cast<const pdb_track_header*>(blob);
        surfaces = reinterpret_cast<const pdb_surface*>(blob + h->offSurfaces);
        tris = reinterpret_cast<const float*>(blob + h->offTris);
        fat = reinterpret_cast<const float*>(blob + h->offFat);
        fatDist = reinterpret_cast<const float*>(blob + h->offFatDist);
        nodes = reinterpret_cast<const float*>(blob + h->offNodes);
        nodeDist = reinterpret_cast<const float*>(blob + h->offNodeDist);
        pits = reinterpret_cast<const float*>(blob + h->offPits);
    }
};

struct RayHitH { bool has = false; float depth = -1; float pos[3] = {0, 0, 0}; float normal[3] = {0, 0, 0}; int surface = -1; };
RayHitH rayCastTrack(const TrackView& tv, const float* origin, const float* dir, float maxDist);

// initial dynamic state of a freshly created car teleported to the spline start
// (Car::Car / Car::init defaults followed by Car::teleportToSpline(0), Car.cpp:1240-1340)
void initialState(const pdb_car_params& P, const TrackView& tv, pdb_dyn_state& S);
// the state edits of Car::teleportToSpline(distanceNorm) applied to an existing state
void teleportToSpline(const pdb_car_params& P, const TrackView& tv, float distanceNorm, pdb_dyn_state& S);
// Car::teleportToPits(pitId) (nothing for an id outside pits.ini's list) and Car::forcePosition(pos) (teleportCarToLocation)
void teleportToPit(const pdb_car_params& P, const TrackView& tv, int pitId, pdb_dyn_state& S);
void teleportToLocation(const pdb_car_params& P, const TrackView& tv, const float* pos, pdb_dyn_state& S);
// Car::teleportByMode: 0 Start, 1 Nearest (trackLocation), 2 Random (the car's own rand() state)
void teleportByMode(const pdb_car_params& P, const TrackView& tv, int mode, pdb_dyn_state& S);

}  // namespace pdb
