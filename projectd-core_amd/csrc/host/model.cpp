// pdbatch host side: builds the per-model constant block (pdb_car_params) from a car's data folder,
// i.e. the load-time half of the reference's component init() routines evaluated once at the
// identity pose (the reference creates every body at the origin with identity rotation and only
// then teleports the car):
//   Car::init / initCarData / initProbes / initLookAhead      Car/Car.cpp:31-314
//   SuspensionStrut::init / attach / setPositions             Car/SuspensionStrut.cpp:17-228
//   SuspensionAxle::init / attach                             Car/SuspensionAxle.cpp:15-118
//   Tyre::initCompounds / setCompound                         Car/Tyre.cpp:48-391
//   TyreThermalModel::buildTyre                               Car/TyreThermalModel.cpp:31-58
//   Drivetrain::init, Engine::init(+precalculate)             Car/Drivetrain.cpp:18-152, Car/Engine.cpp:16-191
//   AutoClutch/AutoBlip/AutoShifter::init                     Car/AutoClutch.cpp:25-89, AutoBlip.cpp:14-49, AutoShifter.cpp:14-29
//   BrakeSystem::init, AeroMap::init, Wing::init              Car/BrakeSystem.cpp:14-73, AeroMap.cpp:15-81, Wing.cpp:19-69
//   Simulator::init                                           Sim/Simulator.cpp:23-87
// Joint anchors follow the ODE setters the reference calls (Physics/ODE/JointODE.cpp:21-60).
// Supported: STRUT / DWB / ML front with AXLE / DWB / ML rear, RWD or FWD, turbochargers, up to 6 wings, heave springs; anything
// else in a car's data (dynamic controllers, EBB, 4WD ...) raises an error naming it (no silent fallback).
#include "model.hpp"
#include "ini.hpp"
#include "rbmath.hpp"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <cfloat>
#include <algorithm>
#include <sys/stat.h>

namespace pdb {

static inline void v3set(float* o, float x, float y, float z) { o[0] = x; o[1] = y; o[2] = z; }
static inline void v3add(float* o, const float* a, const float* b) { o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; }
static inline void v3sub(float* o, const float* a, const float* b) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static inline float v3len(const float* a) { return sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
// vec3f::get_norm() (Core/Math.h:121-125)
static inline void v3norm(float* a) { const float l = v3len(a); if (l != 0.0f) { const float s = 1.0f / l; a[0] *= s; a[1] *= s; a[2] *= s; } }

static bool fileExists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }

struct RawJoint { pdb_joint_def d; };

static void mkDBall(std::vector<RawJoint>& out, const HBody* B, int b0, int b1, const float* p1w, const float* p2w, float erp, float cfm) {
    RawJoint j; memset(&j, 0, sizeof(j));
    j.d.type = PDB_JOINT_DBALL; j.d.b0 = b0; j.d.b1 = b1; j.d.erp = erp; j.d.cfm = cfm; j.d.steerWheel = -1; j.d.suspErp = 1;
    hWorldToLocal(B[b0], p1w, j.d.anchor1);
    hWorldToLocal(B[b1], p2w, j.d.anchor2);
    float g1[3], g2[3], d[3];
    hLocalToWorld(B[b0], j.d.anchor1, g1);
    hLocalToWorld(B[b1], j.d.anchor2, g2);
    v3sub(d, g1, g2);
    j.d.distance = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    j.d.qrel[0] = 1;
    out.push_back(j);
}

// DynamicController::init (Car/DynamicController.cpp:21-115): the stages of one controller file appended to the car's stage table
static void dynCtrlLoad(pdb_car_params& P, pdb_dyn_ctrl& dc, const std::string& path) {
    Ini ini(path);
    dc.first = P.numCtrlStages; dc.count = 0;
    static const char* known[] = {"", "BRAKE", "GAS", "LATG", "LONG", "STEER", "SPEED_KMH", "GEAR", "RPMS", "CONST", "SLIPRATIO_MAX", "SLIPRATIO_AVG", "SLIPANGLE_FRONT_AVG",
                                  "SLIPANGLE_REAR_AVG", "SLIPANGLE_FRONT_MAX", "SLIPANGLE_REAR_MAX", "OVERSTEER_FACTOR", "REAR_SPEED_RATIO", "STEER_DEG", "WHEEL_STEER_DEG",
                                  "LOAD_SPREAD_LF", "LOAD_SPREAD_RF", "AVG_TRAVEL_REAR", "SUS_TRAVEL_LR", "SUS_TRAVEL_RR"};
    for (int id = 0;; ++id) {
        char secn[32]; snprintf(secn, sizeof(secn), "CONTROLLER_%d", id);
        if (!ini.hasSection(secn)) break;
        const std::string in = ini.getString(secn, "INPUT"), comb = ini.getString(secn, "COMBINATOR");
        int iv = 0;
        for (int k = 1; k <= 24; ++k) if (in == known[k]) iv = k;
        if (iv == 0) throw std::runtime_error("pdb: " + path + " " + secn + ": controller input " + in + " unsupported");
        const int cm = comb == "ADD" ? 1 : comb == "MULT" ? 2 : 0;
        if (cm == 0) continue;   // (the reference warns and skips the stage)
        if (P.numCtrlStages >= PDB_MAX_CTRL_STAGES) throw std::runtime_error("pdb: more than " + std::to_string(PDB_MAX_CTRL_STAGES) + " dynamic-controller stages in the car");
        pdb_ctrl_stage& st = P.ctrlStages[P.numCtrlStages++];
        memset(&st, 0, sizeof(st));
        st.input = iv; st.combinator = cm;
        const float lag = ini.getFloat(secn, "FILTER"), orgdt = 0.004f, dt = 0.003f;
        st.filter = (((1.0f / dt) * orgdt) * (1.0f - lag)) * (1.0f / dt);   // lagToLerpDeltaK (DynamicController.cpp:8)
        st.upLimit = ini.getFloat(secn, "UP_LIMIT");
        st.downLimit = ini.getFloat(secn, "DOWN_LIMIT");
        if (iv == 9) st.constValue = ini.getFloat(secn, "CONST_VALUE");
        else {
            const std::string v = ini.getString(secn, "LUT");
            if (v.find(".lut") != std::string::npos) curveLoad(st.lut, path.substr(0, path.find_last_of('/') + 1) + v); else curveParseInline(st.lut, v);
        }
        ++dc.count;
    }
    // a present file without one usable stage: the reference's controller object exists and eval() gives 0 every tick (e.g. diffPreLoad = 0 and
    // diffPowerRamp = 0, Drivetrain.cpp:603-607), while the kernels gate their call sites on count != 0 -- refused rather than stepped differently
    if (dc.count == 0) throw std::runtime_error("pdb: " + path + ": no usable [CONTROLLER_n] stage (the reference would evaluate it to 0 every tick; not supported)");
}

// Natural cubic spline through a LUT (what Curve::getCubicSplineValue evaluates, Core/Curve.cpp:117-126): the reference hands the points to the
// tk::spline header it vendors (Core/tkspline.h, tkfloat = float): second derivative zero at both ends, the tridiagonal system for the b's solved
// by an LU decomposition with rows scaled to a unit diagonal first, a and c from the b's, quadratic continuation past the ends.  Restated here for
// the band width of one, in that header's order of operations and its mixture of float storage and double literals (1.0/3.0, 2.0/3.0, 3.0): the
// coefficients have to come out with the same bits, because the values the tyre model reads from them do.
static void splineBuild(pdb_spline& sp, const pdb_curve& cv) {
    const int n = cv.n;
    if (n < 3 || n > PDB_MAX_SPLINE) throw std::runtime_error("pdb: a cubic-spline tyre LUT needs 3.." + std::to_string(PDB_MAX_SPLINE) + " points");
    for (int i = 0; i + 1 < n; ++i) if (!(cv.x[i] < cv.x[i + 1])) throw std::runtime_error("pdb: cubic-spline tyre LUT: references must increase");
    memset(&sp, 0, sizeof(sp));
    sp.n = n;
    const float* x = cv.x; const float* y = cv.y;
    std::vector<float> lo(n, 0.0f), di(n, 0.0f), up(n, 0.0f), rhs(n, 0.0f), sd(n, 0.0f);   // A(i,i-1), A(i,i), A(i,i+1), right-hand side, 1/diagonal
    for (int i = 1; i < n - 1; ++i) {
        lo[i] = (float)(1.0 / 3.0 * (double)(x[i] - x[i - 1]));
        di[i] = (float)(2.0 / 3.0 * (double)(x[i + 1] - x[i - 1]));
        up[i] = (float)(1.0 / 3.0 * (double)(x[i + 1] - x[i]));
        rhs[i] = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - (y[i] - y[i - 1]) / (x[i] - x[i - 1]);
    }
    di[0] = 2.0f; up[0] = 0.0f; rhs[0] = 0.0f;                 // 2 b[0] = f'' = 0
    di[n - 1] = 2.0f; lo[n - 1] = 0.0f; rhs[n - 1] = 0.0f;
    // rows scaled to a unit diagonal
    for (int i = 0; i < n; ++i) {
        sd[i] = (float)(1.0 / (double)di[i]);
        if (i > 0) lo[i] *= sd[i];
        di[i] *= sd[i];
        if (i < n - 1) up[i] *= sd[i];
        di[i] = 1.0f;
    }
    // elimination
    for (int k = 0; k + 1 < n; ++k) {
        const float f = -lo[k + 1] / di[k];
        lo[k + 1] = -f;
        di[k + 1] = di[k + 1] + f * up[k];
    }
    // L y = b (with the scaling), R x = y
    std::vector<float> yv(n), b(n);
    for (int i = 0; i < n; ++i) {
        float sum = 0.0f;
        if (i > 0) sum += lo[i] * yv[i - 1];
        yv[i] = (rhs[i] * sd[i]) - sum;
    }
    for (int i = n - 1; i >= 0; --i) {
        float sum = 0.0f;
        if (i < n - 1) sum += up[i] * b[i + 1];
        b[i] = (yv[i] - sum) / di[i];
    }
    std::vector<float> a(n, 0.0f), c(n, 0.0f);
    for (int i = 0; i < n - 1; ++i) {
        a[i] = (float)(1.0 / 3.0 * (double)(b[i + 1] - b[i]) / (double)(x[i + 1] - x[i]));
        c[i] = (float)((double)((y[i + 1] - y[i]) / (x[i + 1] - x[i])) - 1.0 / 3.0 * (2.0 * (double)b[i] + (double)b[i + 1]) * (double)(x[i + 1] - x[i]));
    }
    sp.b0 = b[0]; sp.c0 = c[0];
    const float h = x[n - 1] - x[n - 2];
    a[n - 1] = 0.0f;
    c[n - 1] = (float)(3.0 * (double)a[n - 2] * (double)h * (double)h + 2.0 * (double)b[n - 2] * (double)h + (double)c[n - 2]);
    for (int i = 0; i < n; ++i) { sp.x[i] = x[i]; sp.y[i] = y[i]; sp.a[i] = a[i]; sp.b[i] = b[i]; sp.c[i] = c[i]; }
}

static void loadTyre(pdb_tyre& t, pdb_car_params& P, const std::string& dataPath, int index) {
    Ini ini(dataPath + "tyres.ini");
    if (!ini.ready) throw std::runtime_error("pdb: tyres.ini not found in " + dataPath);
    const int iVer = ini.getInt("HEADER", "VERSION");
    if (iVer < 10) throw std::runtime_error("pdb: tyres.ini VERSION < 10 unsupported (Tyre.cpp:54)");
    const std::string sec = (index < 2) ? "FRONT" : "REAR";
    memset(&t, 0, sizeof(t));
    t.version = iVer;
    // defaults (Car/Tyre.h:62-70, Car/TyreCompound.h)
    t.flatSpotK = 0.15f; t.explosionTemperature = 350.0f; t.pressureTemperatureGain = 0.16f;
    t.camberSpreadK = 1.4f;
    t.surfaceTransfer = 0.3f; t.patchTransfer = 0.2f; t.patchCoreTransfer = 0.2f; t.internalCoreTransfer = 0.004f; t.coolFactorGain = 0;
    t.thermalFrictionK = 0.03f; t.thermalRollingK = 0.5f; t.thermalRollingSurfaceK = 0;
    t.dCamberBlend = 1.0f; t.brakeDXMod = 1.0f; t.combinedFactor = 0.0f; t.cfXmult = 1.0f;
    if (ini.hasSection("EXPLOSION")) t.explosionTemperature = ini.getFloat("EXPLOSION", "TEMPERATURE");
    if (ini.hasSection("ADDITIONAL1")) {
        t.pressureTemperatureGain = ini.getFloat("ADDITIONAL1", "PRESSURE_TEMPERATURE_GAIN");
        const float sp = ini.getFloat("ADDITIONAL1", "CAMBER_TEMP_SPREAD_K");
        if (sp != 0.0f) t.camberSpreadK = sp;
    }
    // DY_CURVE / DX_CURVE / DCAMBER_LUT (Tyre.cpp:161-165,207-211): LUTs read through a natural cubic spline (Curve::getCubicSplineValue)
    t.curveFlags = 0;
    {
        auto lutOf = [&](const char* key, pdb_spline& sp) {
            pdb_curve cv; memset(&cv, 0, sizeof(cv));
            const std::string v = ini.getString(sec, key);
            if (v.find(".lut") != std::string::npos) curveLoad(cv, dataPath + v); else curveParseInline(cv, v);   // INIReader::getCurve (INIReader.cpp:232-253)
            if (cv.n > 0) splineBuild(sp, cv);
            return cv.n > 0;
        };
        if (ini.hasKey(sec, "DY_CURVE") && lutOf("DY_CURVE", t.dyLoadCurve)) t.curveFlags |= 1;
        if (ini.hasKey(sec, "DX_CURVE") && lutOf("DX_CURVE", t.dxLoadCurve)) t.curveFlags |= 2;
        if (ini.hasKey(sec, "DCAMBER_LUT") && lutOf("DCAMBER_LUT", t.dCamberCurve)) {
            t.curveFlags |= 4;
            if (ini.getInt(sec, "DCAMBER_LUT_SMOOTH") != 0) t.curveFlags |= 8;
        }
    }
    float width = ini.getFloat(sec, "WIDTH"); (void)width;
    t.radius = ini.getFloat(sec, "RADIUS");
    t.rimRadius = ini.getFloat(sec, "RIM_RADIUS");
    float fFLA = ini.getFloat(sec, "FRICTION_LIMIT_ANGLE");
    if (fFLA == 0.0f) fFLA = 7.5f;
    t.cfXmult = ini.getFloat(sec, "CX_MULT");
    t.radiusRaiseK = ini.getFloat(sec, "RADIUS_ANGULAR_K") * 0.001f;
    if (ini.hasKey(sec, "BRAKE_DX_MOD")) {
        t.brakeDXMod = ini.getFloat(sec, "BRAKE_DX_MOD");
        if (t.brakeDXMod == 0.0f) t.brakeDXMod = 1.0f; else t.brakeDXMod += 1.0f;
    }
    if (ini.hasKey(sec, "COMBINED_FACTOR")) t.combinedFactor = ini.getFloat(sec, "COMBINED_FACTOR");
    const float fFZ0 = ini.getFloat(sec, "FZ0");
    const float fFlexGain = ini.getFloat(sec, "FLEX_GAIN");
    t.lsExpX = ini.getFloat(sec, "LS_EXPX");
    t.lsExpY = ini.getFloat(sec, "LS_EXPY");
    float Dx0 = ini.getFloat(sec, "DX_REF");
    float Dy0 = ini.getFloat(sec, "DY_REF");
    t.lsMultX = (Dx0 * fFZ0) / powf(fFZ0, t.lsExpX);   // calcLoadSensMult (TyreUtils.inl:32-35)
    t.lsMultY = (Dy0 * fFZ0) / powf(fFZ0, t.lsExpY);
    t.Fz0 = fFZ0;
    t.modelFz0 = 2000.0f;
    t.maxSlip0 = tanf(fFLA * 0.017453f);
    t.maxSlip1 = tanf(((fFlexGain + 1.0f) * fFLA) * 0.017453f);
    t.asy = ini.getFloat(sec, "FALLOFF_LEVEL");
    t.falloffSpeed = ini.getFloat(sec, "FALLOFF_SPEED");
    t.speedSensitivity = ini.getFloat(sec, "SPEED_SENSITIVITY");
    t.relaxationLength = ini.getFloat(sec, "RELAXATION_LENGTH");
    t.rr0 = ini.getFloat(sec, "ROLLING_RESISTANCE_0");
    t.rr1 = ini.getFloat(sec, "ROLLING_RESISTANCE_1");
    t.rr_slip = ini.getFloat(sec, "ROLLING_RESISTANCE_SLIP");
    t.camberGain = ini.getFloat(sec, "CAMBER_GAIN");
    t.dcamber0 = ini.getFloat(sec, "DCAMBER_0");
    t.dcamber1 = ini.getFloat(sec, "DCAMBER_1");
    if (t.dcamber0 == 0.0f || t.dcamber1 == 0.0f) { t.dcamber0 = 0.1f; t.dcamber1 = -0.8f; }
    t.angularInertia = ini.getFloat(sec, "ANGULAR_INERTIA");
    t.d = ini.getFloat(sec, "DAMP");
    t.k = ini.getFloat(sec, "RATE");
    if (t.angularInertia == 0.0f) t.angularInertia = 1.2f;
    if (t.d == 0.0f) t.d = 400.0f;
    if (t.k == 0.0f) t.k = 220000.0f;
    t.pressureStatic = ini.getFloat(sec, "PRESSURE_STATIC");
    if (t.pressureStatic == 0.0f) t.pressureStatic = 26.0f;
    t.pressureRef = t.pressureStatic;
    t.pressureSpringGain = ini.getFloat(sec, "PRESSURE_SPRING_GAIN");
    if (t.pressureSpringGain == 0.0f) t.pressureSpringGain = 1000.0f;
    t.pressureCfGain = ini.getFloat(sec, "PRESSURE_FLEX_GAIN");
    t.pressureRRGain = ini.getFloat(sec, "PRESSURE_RR_GAIN");
    t.pressureGainD = ini.getFloat(sec, "PRESSURE_D_GAIN");
    t.idealPressure = ini.getFloat(sec, "PRESSURE_IDEAL");
    if (t.idealPressure == 0.0f) t.idealPressure = 26.0f;
    const std::string th = "THERMAL_" + sec;
    if (ini.hasSection(th)) {
        t.surfaceTransfer = ini.getFloat(th, "SURFACE_TRANSFER");
        t.patchTransfer = ini.getFloat(th, "PATCH_TRANSFER");
        t.patchCoreTransfer = ini.getFloat(th, "CORE_TRANSFER");
        t.thermalFrictionK = ini.getFloat(th, "FRICTION_K");
        t.thermalRollingK = ini.getFloat(th, "ROLLING_K");
        t.internalCoreTransfer = ini.getFloat(th, "INTERNAL_CORE_TRANSFER");
        if (ini.hasKey(th, "COOL_FACTOR")) t.coolFactorGain = (ini.getFloat(th, "COOL_FACTOR") - 1.0f) * 0.000324f;
        t.thermalRollingSurfaceK = ini.getFloat(th, "SURFACE_ROLLING_K");
        curveLoad(t.performanceCurve, dataPath + ini.getString(th, "PERFORMANCE_CURVE"));
    }
    curveLoad(t.wearCurve, dataPath + ini.getString(sec, "WEAR_CURVE"));
    for (int i = 0; i < t.wearCurve.n; ++i) t.wearCurve.y[i] *= 0.01f;
    // softnessIndex = max(0, loadSensExpD(lsExpY, lsMultY, 3000) - 1)  (Tyre.cpp:323-329)
    const float sens = (powf(3000.0f, t.lsExpY) * t.lsMultY) / 3000.0f;
    t.softnessIndex = std::max(0.0f, sens - 1.0f);
    (void)P;
}

static void buildPatchConn(pdb_car_params& P) {
    // Car/TyreThermalModel.cpp:31-58 with stripes = 3, elements = 12; patch index = y + x*elements
    const int stripes = 3, elements = 12;
    for (int k = 0; k < 36; ++k) P.patchConnCount[k] = 0;
    auto connect = [&](int a, int b) {
        P.patchConn[a][(int)P.patchConnCount[a]++] = (int8_t)b;
        P.patchConn[b][(int)P.patchConnCount[b]++] = (int8_t)a;
    };
    for (int i = 0; i < stripes; ++i)
        for (int j = 0; j < elements; ++j) {
            const int p = j + i * elements;
            if (i + 1 < stripes) connect(p, j + (i + 1) * elements);
            if (j + 1 < elements) connect(p, (j + 1) + i * elements);
            else if (j + 1 == elements) connect(p, 0 + i * elements);
        }
}

// ODE island traversal order (util.cpp dxProcessIslands) -> solver row order
static std::vector<int> islandOrder(const std::vector<RawJoint>& J, int nb) {
    std::vector<std::vector<int>> bj(nb);
    for (int j = 0; j < (int)J.size(); ++j) {
        bj[J[j].d.b0].insert(bj[J[j].d.b0].begin(), j);
        bj[J[j].d.b1].insert(bj[J[j].d.b1].begin(), j);
    }
    std::vector<char> bt(nb, 0), jt(J.size(), 0);
    std::vector<int> order, stack;
    for (int bb = nb - 1; bb >= 0; --bb) {
        if (bt[bb]) continue;
        bt[bb] = 1;
        int b = bb;
        stack.clear();
        for (;;) {
            for (int j : bj[b]) {
                if (jt[j]) continue;
                jt[j] = 1;
                order.push_back(j);
                const int other = (J[j].d.b0 == b) ? J[j].d.b1 : J[j].d.b0;
                if (!bt[other]) { bt[other] = 1; stack.push_back(other); }
            }
            if (stack.empty()) break;
            b = stack.back();
            stack.pop_back();
        }
    }
    return order;
}

void buildCarModel(const std::string& basePathIn, const std::string& modelName, pdb_car_params& P) {
    std::string base = basePathIn;
    std::replace(base.begin(), base.end(), '\\', '/');
    if (!base.empty() && base.back() != '/') base += '/';
    const std::string dataPath = base + "content/cars/" + modelName + "/data/";
    {   // packed block (the product's own distribution format for a configured car): <model>/<model>.pdcar next to data/
        FILE* probe = fopen((dataPath + "car.ini").c_str(), "rb");
        if (probe) fclose(probe);
        else if (FILE* f = fopen((base + "content/cars/" + modelName + "/" + modelName + ".pdcar").c_str(), "rb")) {
            const size_t got = fread(&P, 1, sizeof(P), f);
            const bool tail = fgetc(f) != EOF;
            fclose(f);
            if (got != sizeof(P) || tail || P.magic != 0x50434450 || P.version != 1) throw std::runtime_error("pdb: malformed packed car block for " + modelName);
            return;
        }
    }
    memset(&P, 0, sizeof(P));
    P.magic = 0x50434450;  // 'PDCP'
    P.version = 1;

    // ---- Simulator::init (Sim/Simulator.cpp:33-70, 346-349) ----
    P.roadTemperature = 20.0f; P.ambientTemperature = 20.0f;
    P.fuelConsumptionRate = 0.0f; P.tyreConsumptionRate = 0.0f; P.mechanicalDamageRate = 1.0f;
    Ini simIni(base + "cfg/sim.ini");
    if (simIni.ready) {
        simIni.tryGetFloat("ENVIRONMENT", "ROAD_TEMP", P.roadTemperature);
        simIni.tryGetFloat("ENVIRONMENT", "AMBIENT_TEMP", P.ambientTemperature);
    }
    P.airDensity = 1.2922f - (P.ambientTemperature * 0.0041f);
    P.worldErp = 0.3f; P.worldCfm = 1.0e-7f;               // PhysicsEngineODE.cpp:24-25
    v3set(P.gravity, 0.0f, -9.80665f, 0.0f);                // :23

    // ---- Car::initCarData ----
    Ini car(dataPath + "car.ini");
    if (!car.ready) throw std::runtime_error("pdb: car.ini not found: " + dataPath);
    if (car.hasSection("EXPLICIT_INERTIA")) throw std::runtime_error("pdb: EXPLICIT_INERTIA cars unsupported this round");
    P.mass = car.getFloat("BASIC", "TOTALMASS");
    float bodyInertia[3];
    car.getFloat3("BASIC", "INERTIA", bodyInertia);
    P.fuelKG = 0.74f;
    if (car.hasSection("FUEL_EXT")) P.fuelKG = car.getFloat("FUEL_EXT", "KG_PER_LITER");
    P.steerLock = car.getFloat("CONTROLS", "STEER_LOCK");
    P.steerRatio = car.getFloat("CONTROLS", "STEER_RATIO");
    P.steerLinearRatio = car.getFloat("CONTROLS", "LINEAR_STEER_ROD_RATIO");
    if (P.steerLinearRatio == 0.0f) P.steerLinearRatio = 0.003f;
    P.fuelConsumptionK = car.getFloat("FUEL", "CONSUMPTION");
    P.fuel = car.getFloat("FUEL", "FUEL");
    if (P.fuel == 0.0f) P.fuel = 30.0f;
    car.getFloat3("FUELTANK", "POSITION", P.fuelTankPos);

    // ---- probes / look-ahead (Car.cpp:285-314) ----
    {
        int n = 0;
        for (int id = 1; id <= 10 && simIni.ready; ++id) {
            char secn[32]; snprintf(secn, sizeof(secn), "CAR_PROBE_%d", id);
            if (!simIni.hasSection(secn)) break;
            if (n >= PDB_NUM_PROBES) throw std::runtime_error("pdb: more than 7 probes unsupported");
            const float yaw = simIni.getFloat(secn, "YAW");
            P.probeLen[n] = simIni.getFloat(secn, "LENGTH");
            const float ax[3] = {0, 1, 0};
            float M[9];
            hAxisAngle(ax, yaw * (float)0.01745329251994329576923690768489, M);
            // vec3f(0,0,1) * mat44f  (Core/Math.h:196-203): out.c = M4c + dot(v, column c)
            const float v[3] = {0, 0, 1};
            P.probeDir[n][0] = 0.0f + (v[0] * M[0] + v[1] * M[3] + v[2] * M[6]);
            P.probeDir[n][1] = 0.0f + (v[0] * M[1] + v[1] * M[4] + v[2] * M[7]);
            P.probeDir[n][2] = 0.0f + (v[0] * M[2] + v[1] * M[5] + v[2] * M[8]);
            ++n;
        }
        if (n != PDB_NUM_PROBES) throw std::runtime_error("pdb: cfg/sim.ini must define exactly 7 CAR_PROBE_n sections");
        P.lookAheadCount = 5; P.lookAheadStep = 10.0f;
        if (simIni.ready) { simIni.tryGetInt("CAR_LOOK_AHEAD", "COUNT", P.lookAheadCount); simIni.tryGetFloat("CAR_LOOK_AHEAD", "STEP", P.lookAheadStep); }
        if (P.lookAheadCount != PDB_NUM_LOOKAHEAD) throw std::runtime_error("pdb: CAR_LOOK_AHEAD COUNT must be 5");
    }

    // ---- bodies at the identity pose ----
    HBody B[PDB_MAX_BODIES];
    hBoxInertia(P.mass, bodyInertia[0], bodyInertia[1], bodyInertia[2], B[0].inertia); B[0].mass = P.mass;
    hBoxInertia(1.0f, 0.5f, 0.5f, 0.5f, B[1].inertia); B[1].mass = 1.0f;   // Car.cpp:49
    memcpy(B[1].pos, P.fuelTankPos, sizeof(float) * 3);                     // Car.cpp:50
    std::vector<RawJoint> J;
    {   // fixed joint (tank, chassis)  Car.cpp:51, JointODE.cpp:21-28
        RawJoint j; memset(&j, 0, sizeof(j));
        j.d.type = PDB_JOINT_FIXED; j.d.b0 = PDB_BODY_TANK; j.d.b1 = PDB_BODY_CHASSIS; j.d.erp = P.worldErp; j.d.cfm = P.worldCfm; j.d.steerWheel = -1;
        float ofs[3]; v3sub(ofs, B[1].pos, B[0].pos);
        hMul1(j.d.offset, B[1].R, ofs);
        hQMul1(j.d.qrel, B[1].q, B[0].q);
        J.push_back(j);
    }

    Ini sus(dataPath + "suspensions.ini");
    if (!sus.ready) throw std::runtime_error("pdb: suspensions.ini not found");
    const std::string typeF = sus.getString("FRONT", "TYPE"), typeR = sus.getString("REAR", "TYPE");
    if ((typeF != "STRUT" && typeF != "DWB" && typeF != "ML") || (typeR != "AXLE" && typeR != "DWB" && typeR != "ML"))
        throw std::runtime_error("pdb: suspension types " + typeF + "/" + typeR + " are not implemented (front STRUT|DWB|ML, rear AXLE|DWB|ML)");
    P.suspTypeF = (typeF == "STRUT") ? PDB_SUSP_STRUT : (typeF == "ML") ? PDB_SUSP_ML : PDB_SUSP_DW;
    P.suspTypeR = (typeR == "AXLE") ? PDB_SUSP_AXLE : (typeR == "ML") ? PDB_SUSP_ML : PDB_SUSP_DW;
    if (typeR == "AXLE") P.axleTorqueReaction = sus.getFloat("AXLE", "TORQUE_REACTION");
    // body slots in the reference's creation order (Car.cpp:38-39,63-107): chassis, tank, [rigid axle], then per wheel
    // hub (+ strut body for struts)
    int nextBody = 2;
    const int axleB = (typeR == "AXLE") ? nextBody++ : -1;
    int hubOf[4], strutOf[4];
    for (int index = 0; index < 4; ++index) {
        const bool front = index < 2;
        const std::string& ty = front ? typeF : typeR;
        strutOf[index] = -1;
        if (ty == "AXLE") { hubOf[index] = axleB; continue; }
        hubOf[index] = nextBody++;
        if (ty == "STRUT") strutOf[index] = nextBody++;
    }
    if (nextBody > PDB_MAX_BODIES) throw std::runtime_error("pdb: too many rigid bodies for this suspension combination");
    const int iVer = sus.getInt("HEADER", "VERSION");
    const float wheelBase = sus.getFloat("BASIC", "WHEELBASE");
    const float cg = sus.getFloat("BASIC", "CG_LOCATION");
    const float frontBaseY = sus.getFloat("FRONT", "BASEY");
    const float frontTrack = sus.getFloat("FRONT", "TRACK") * 0.5f;
    const float rearBaseY = sus.getFloat("REAR", "BASEY");
    const float rearTrack = sus.getFloat("REAR", "TRACK") * 0.5f;

    auto loadDamper = [&](pdb_damper& d, const std::string& s) {
        d.bumpSlow = sus.getFloat(s, "DAMP_BUMP"); d.reboundSlow = sus.getFloat(s, "DAMP_REBOUND");
        d.bumpFast = sus.getFloat(s, "DAMP_FAST_BUMP"); d.reboundFast = sus.getFloat(s, "DAMP_FAST_REBOUND");
        d.fastThresholdBump = sus.getFloat(s, "DAMP_FAST_BUMPTHRESHOLD"); d.fastThresholdRebound = sus.getFloat(s, "DAMP_FAST_REBOUNDTHRESHOLD");
        if (d.fastThresholdBump == 0.0f) d.fastThresholdBump = 0.2f;
        if (d.fastThresholdRebound == 0.0f) d.fastThresholdRebound = 0.2f;
        if (d.bumpFast == 0.0f) d.bumpFast = d.bumpSlow;
        if (d.reboundFast == 0.0f) d.reboundFast = d.reboundSlow;
    };

    // ---- double wishbone (SuspensionDW.cpp:17-207): hub body + 5 distance joints, front or rear ----
    auto initDW = [&](int index) {
        pdb_susp& S = P.susp[index];
        memset(&S, 0, sizeof(S));
        S.type = PDB_SUSP_DW;
        const int hubB = hubOf[index];
        S.hubBody = hubB; S.strutBody = -1;
        const std::string id = (index < 2) ? "FRONT" : "REAR";
        float ref[3];
        if (index < 2) v3set(ref, (index == 0) ? frontTrack : -frontTrack, frontBaseY, (1.0f - cg) * wheelBase);
        else v3set(ref, (index == 2) ? rearTrack : -rearTrack, rearBaseY, -(cg * wheelBase));
        float carTopF[3], carTopR[3], carBotF[3], carBotR[3], tyreTop[3], tyreBot[3], tyreSteer[3], carSteer[3];
        sus.getFloat3(id, "WBCAR_TOP_FRONT", carTopF); sus.getFloat3(id, "WBCAR_TOP_REAR", carTopR);
        sus.getFloat3(id, "WBCAR_BOTTOM_FRONT", carBotF); sus.getFloat3(id, "WBCAR_BOTTOM_REAR", carBotR);
        sus.getFloat3(id, "WBTYRE_TOP", tyreTop); sus.getFloat3(id, "WBTYRE_BOTTOM", tyreBot);
        sus.getFloat3(id, "WBTYRE_STEER", tyreSteer); sus.getFloat3(id, "WBCAR_STEER", carSteer);
        float* all[8] = {carTopF, carTopR, carBotF, carBotR, tyreTop, tyreBot, tyreSteer, carSteer};
        if (iVer >= 2) {
            const float rim = -sus.getFloat(id, "RIM_OFFSET");
            if (rim != 0.0f) for (auto* p : all) p[0] += rim;
        }
        float hubMass = sus.getFloat(id, "HUB_MASS");
        S.bumpStopUp = sus.getFloat(id, "BUMPSTOP_UP");
        S.bumpStopDn = -sus.getFloat(id, "BUMPSTOP_DN");
        S.rodLength = sus.getFloat(id, "ROD_LENGTH");
        S.toeOutLinear = sus.getFloat(id, "TOE_OUT");
        S.k = sus.getFloat(id, "SPRING_RATE");
        S.progressiveK = sus.getFloat(id, "PROGRESSIVE_SPRING_RATE");
        loadDamper(S.damper, id);
        S.bumpStopRate = sus.getFloat(id, "BUMP_STOP_RATE");
        if (S.bumpStopRate == 0.0f) S.bumpStopRate = 500000.0f;
        if (sus.hasKey(id, "BUMP_STOP_PROGRESSIVE")) S.bumpStopProgressive = sus.getFloat(id, "BUMP_STOP_PROGRESSIVE");
        S.staticCamber = -sus.getFloat(id, "STATIC_CAMBER") * 0.017453f;
        if (index % 2) S.staticCamber *= -1.0f;
        S.packerRange = sus.getFloat(id, "PACKER_RANGE");
        if (ref[0] > 0.0f) for (auto* p : all) p[0] *= -1.0f;
        if (hubMass <= 0.0f) hubMass = 20.0f;
        B[hubB].mass = hubMass; hBoxInertia(hubMass, 0.2f, 0.6f, 0.6f, B[hubB].inertia);
        S.mass = hubMass;
        memcpy(S.basePosition, ref, sizeof(ref));
        S.refPointY = ref[1];
        S.refPointSignX = (ref[0] > 0.0f) ? 1.0f : ((ref[0] < 0.0f) ? -1.0f : 0.0f);
        // attach(): hub at the reference point with the body's orientation; dataRelToBody.X = localToWorld(X + refPoint)
        float Mb[9]; hWorldMatrix3(B[0], Mb);
        hSetRotationM(B[hubB], Mb);
        hLocalToWorld(B[0], S.basePosition, B[hubB].pos);
        float t3[3], rbTopF[3], rbTopR[3], rbBotF[3], rbBotR[3], rbTyreTop[3], rbTyreBot[3], rbCarSteer[3], rbTyreSteer[3];
        auto rel = [&](const float* p, float* o) { v3add(t3, p, ref); hLocalToWorld(B[0], t3, o); };
        rel(carBotF, rbBotF); rel(carBotR, rbBotR); rel(carTopF, rbTopF); rel(carTopR, rbTopR);
        rel(tyreBot, rbTyreBot); rel(tyreTop, rbTyreTop); rel(carSteer, rbCarSteer); rel(tyreSteer, rbTyreSteer);
        mkDBall(J, B, PDB_BODY_CHASSIS, hubB, rbTopR, rbTyreTop, P.worldErp, P.worldCfm);
        mkDBall(J, B, PDB_BODY_CHASSIS, hubB, rbTopF, rbTyreTop, P.worldErp, P.worldCfm);
        mkDBall(J, B, PDB_BODY_CHASSIS, hubB, rbBotR, rbTyreBot, P.worldErp, P.worldCfm);
        mkDBall(J, B, PDB_BODY_CHASSIS, hubB, rbBotF, rbTyreBot, P.worldErp, P.worldCfm);
        mkDBall(J, B, PDB_BODY_CHASSIS, hubB, rbCarSteer, rbTyreSteer, P.worldErp, P.worldCfm);
        memcpy(S.tyreSteer, tyreSteer, sizeof(float) * 3);
        memcpy(S.baseCarSteer, rbCarSteer, sizeof(float) * 3);
        // setSteerLengthOffset(0): the steering link reseated with the toe-out offset (front: every tick by SteeringSystem::step,
        // rear: once, here) -- reseatDistanceJointLocal(carSteer + (sign(x) toe, 0, 0), tyreSteer)
        {
            RawJoint& sj = J.back();
            const float offx = 0.0f + 0.0f + (S.refPointSignX * S.toeOutLinear);
            float cs[3] = {rbCarSteer[0] + offx, rbCarSteer[1], rbCarSteer[2]}, w[3];
            hLocalToWorld(B[0], cs, w); hWorldToLocal(B[0], w, sj.d.anchor1);
            hLocalToWorld(B[hubB], tyreSteer, w); hWorldToLocal(B[hubB], w, sj.d.anchor2);
            if (index < 2) sj.d.steerWheel = index;
        }
        for (size_t k = J.size() - 5; k < J.size(); ++k) { J[k].d.erp = 0.3f; J[k].d.cfm = 0.0000001f; }   // setERPCFM(0.3, baseCFM)
    };

    // ---- multilink (SuspensionML.cpp:16-104): hub body + 5 distance joints given as (car, tyre) ball pairs relative to the wheel;
    //      no bump stops, damper thresholds as written in the file, joints keep the world ERP/CFM (setERPCFM is empty) ----
    auto initML = [&](int index) {
        pdb_susp& S = P.susp[index];
        memset(&S, 0, sizeof(S));
        S.type = PDB_SUSP_ML;
        const int hubB = hubOf[index];
        S.hubBody = hubB; S.strutBody = -1;
        const std::string id = (index < 2) ? "FRONT" : "REAR";
        float ref[3];
        if (index < 2) v3set(ref, (index == 0) ? frontTrack : -frontTrack, frontBaseY, (1.0f - cg) * wheelBase);
        else v3set(ref, (index == 2) ? rearTrack : -rearTrack, rearBaseY, -(cg * wheelBase));
        memcpy(S.basePosition, ref, sizeof(ref));
        S.refPointY = ref[1];
        S.refPointSignX = (ref[0] > 0.0f) ? 1.0f : ((ref[0] < 0.0f) ? -1.0f : 0.0f);
        const float hubMass = sus.getFloat(id, "HUB_MASS");
        B[hubB].mass = hubMass; hBoxInertia(hubMass, 0.2f, 0.6f, 0.6f, B[hubB].inertia);
        S.mass = hubMass;
        float Mb[9]; hWorldMatrix3(B[0], Mb);
        hSetRotationM(B[hubB], Mb);
        hLocalToWorld(B[0], S.basePosition, B[hubB].pos);
        for (int i = 0; i < 5; ++i) {
            char kc[32], kt[32]; snprintf(kc, sizeof(kc), "JOINT%d_CAR", i); snprintf(kt, sizeof(kt), "JOINT%d_TYRE", i);
            float car[3], tyre[3], w[3], relCarC[3], relCarT[3], vC[3], vT[3];
            sus.getFloat3(id, kc, car); sus.getFloat3(id, kt, tyre);
            if (ref[0] > 0.0f) { car[0] *= -1.0f; tyre[0] *= -1.0f; }
            hLocalToWorld(B[hubB], car, w); hWorldToLocal(B[0], w, relCarC);     // ballCar.relToCar
            hLocalToWorld(B[hubB], tyre, w); hWorldToLocal(B[0], w, relCarT);    // ballTyre.relToCar
            hLocalToWorld(B[0], relCarC, vC); hLocalToWorld(B[0], relCarT, vT);
            mkDBall(J, B, PDB_BODY_CHASSIS, hubB, vC, vT, P.worldErp, P.worldCfm);
            J.back().d.suspErp = 0;
            if (i == 4) {
                memcpy(S.baseCarSteer, relCarC, sizeof(float) * 3);   // baseCarSteerPosition = joints[4].ballCar.relToCar
                memcpy(S.tyreSteer, tyre, sizeof(float) * 3);         // joints[4].ballTyre.relToTyre
                if (index < 2) J.back().d.steerWheel = index;
            }
        }
        S.rodLength = sus.getFloat(id, "ROD_LENGTH");
        S.toeOutLinear = sus.getFloat(id, "TOE_OUT");
        S.k = sus.getFloat(id, "SPRING_RATE");
        S.progressiveK = sus.getFloat(id, "PROGRESSIVE_SPRING_RATE");
        S.damper.bumpSlow = sus.getFloat(id, "DAMP_BUMP"); S.damper.reboundSlow = sus.getFloat(id, "DAMP_REBOUND");
        S.damper.bumpFast = sus.getFloat(id, "DAMP_FAST_BUMP"); S.damper.reboundFast = sus.getFloat(id, "DAMP_FAST_REBOUND");
        S.damper.fastThresholdBump = sus.getFloat(id, "DAMP_FAST_BUMPTHRESHOLD"); S.damper.fastThresholdRebound = sus.getFloat(id, "DAMP_FAST_REBOUNDTHRESHOLD");
        S.staticCamber = -sus.getFloat(id, "STATIC_CAMBER") * 0.017453f;
        if (index % 2) S.staticCamber *= -1.0f;
        // packerRange / bumpStopRate / bump stops keep the SuspensionBase defaults (0): SuspensionML::init never reads them
    };

    // ---- front struts (SuspensionStrut.cpp:17-228) ----
    for (int index = 0; index < 2; ++index) {
        if (typeF == "DWB") { initDW(index); continue; }
        if (typeF == "ML") { initML(index); continue; }
        pdb_susp& S = P.susp[index];
        memset(&S, 0, sizeof(S));
        S.type = PDB_SUSP_STRUT;
        const int hubB = hubOf[index];
        const int strB = strutOf[index];
        S.hubBody = hubB; S.strutBody = strB;
        const std::string id = "FRONT";
        float ref[3];
        v3set(ref, (index == 0) ? frontTrack : -frontTrack, frontBaseY, (1.0f - cg) * wheelBase);
        float carStrut[3], tyreStrut[3], carWBF[3], carWBR[3], tyreWB[3], tyreSteer[3], carSteer[3];
        sus.getFloat3(id, "STRUT_CAR", carStrut); sus.getFloat3(id, "STRUT_TYRE", tyreStrut);
        sus.getFloat3(id, "WBCAR_BOTTOM_FRONT", carWBF); sus.getFloat3(id, "WBCAR_BOTTOM_REAR", carWBR);
        sus.getFloat3(id, "WBTYRE_BOTTOM", tyreWB); sus.getFloat3(id, "WBTYRE_STEER", tyreSteer); sus.getFloat3(id, "WBCAR_STEER", carSteer);
        float* all[7] = {carStrut, tyreStrut, carWBF, carWBR, tyreWB, tyreSteer, carSteer};
        if (iVer >= 2) {
            const float rim = -sus.getFloat(id, "RIM_OFFSET");
            if (rim != 0.0f) for (auto* p : all) p[0] += rim;
        }
        float hubMass = sus.getFloat(id, "HUB_MASS");
        S.bumpStopUp = sus.getFloat(id, "BUMPSTOP_UP");
        S.bumpStopDn = -sus.getFloat(id, "BUMPSTOP_DN");
        S.rodLength = sus.getFloat(id, "ROD_LENGTH");
        S.toeOutLinear = sus.getFloat(id, "TOE_OUT");
        S.k = sus.getFloat(id, "SPRING_RATE");
        S.progressiveK = sus.getFloat(id, "PROGRESSIVE_SPRING_RATE");
        loadDamper(S.damper, id);
        S.bumpStopRate = sus.getFloat(id, "BUMP_STOP_RATE");
        if (S.bumpStopRate == 0.0f) S.bumpStopRate = 500000.0f;
        S.staticCamber = -sus.getFloat(id, "STATIC_CAMBER") * 0.017453f;
        if (index % 2) S.staticCamber *= -1.0f;
        S.packerRange = sus.getFloat(id, "PACKER_RANGE");
        if (ref[0] > 0.0f) for (auto* p : all) p[0] *= -1.0f;
        if (hubMass <= 0.0f) hubMass = 20.0f;
        B[hubB].mass = hubMass * 0.8f; hBoxInertia(B[hubB].mass, 0.2f, 0.6f, 0.6f, B[hubB].inertia);
        B[strB].mass = hubMass * 0.2f; hBoxInertia(B[strB].mass, 0.05f, 0.5f, 0.2f, B[strB].inertia);
        S.strutBodyLength = 0.2f;
        S.mass = B[hubB].mass;
        memcpy(S.basePosition, ref, sizeof(ref));
        S.refPointY = ref[1];
        S.refPointSignX = (ref[0] > 0.0f) ? 1.0f : ((ref[0] < 0.0f) ? -1.0f : 0.0f);
        // attach(): dataRelToBody.X = carBody->localToWorld(dataRelToWheel.X + refPoint), body at identity
        float rb_carWBF[3], rb_carWBR[3], rb_carStrut[3], rb_tyreWB[3], rb_tyreStrut[3], rb_carSteer[3], rb_tyreSteer[3], t3[3];
        auto rel = [&](const float* p, float* o) { v3add(t3, p, ref); hLocalToWorld(B[0], t3, o); };
        rel(carWBF, rb_carWBF); rel(carWBR, rb_carWBR); rel(carStrut, rb_carStrut); rel(tyreWB, rb_tyreWB);
        rel(tyreStrut, rb_tyreStrut); rel(carSteer, rb_carSteer); rel(tyreSteer, rb_tyreSteer);
        // setPositions()
        float Mb[9]; hWorldMatrix3(B[0], Mb);
        float vPos[3]; hLocalToWorld(B[0], S.basePosition, vPos);
        hSetRotationM(B[hubB], Mb);
        memcpy(B[hubB].pos, vPos, sizeof(vPos));
        float vCarStrut[3], vTyreStrut[3], vNorm[3];
        hLocalToWorld(B[0], rb_carStrut, vCarStrut);
        hLocalToWorld(B[hubB], tyreStrut, vTyreStrut);
        v3sub(vNorm, vTyreStrut, vCarStrut); v3norm(vNorm);
        float vM3[3] = {Mb[6] * -1.0f, Mb[7] * -1.0f, Mb[8] * -1.0f};
        float vM3N[3], vM3NN[3];
        hCross(vM3N, vM3, vNorm);
        hCross(vM3NN, vM3N, vNorm); v3norm(vM3NN);
        const float Ms[9] = {vM3NN[0], vM3NN[1], vM3NN[2], -vM3N[0], -vM3N[1], -vM3N[2], -vNorm[0], -vNorm[1], -vNorm[2]};
        hSetRotationM(B[strB], Ms);
        B[strB].pos[0] = (vNorm[0] * S.strutBodyLength) * 0.5f + vCarStrut[0];
        B[strB].pos[1] = (vNorm[1] * S.strutBodyLength) * 0.5f + vCarStrut[1];
        B[strB].pos[2] = (vNorm[2] * S.strutBodyLength) * 0.5f + vCarStrut[2];
        // joints 0..2 distance, 3 slider, 4 ball
        mkDBall(J, B, PDB_BODY_CHASSIS, hubB, rb_carWBR, rb_tyreWB, P.worldErp, P.worldCfm);
        mkDBall(J, B, PDB_BODY_CHASSIS, hubB, rb_carWBF, rb_tyreWB, P.worldErp, P.worldCfm);
        mkDBall(J, B, PDB_BODY_CHASSIS, hubB, rb_carSteer, rb_tyreSteer, P.worldErp, P.worldCfm);
        J.back().d.steerWheel = index;
        {   // slider (strut, hub) along (vTyreStrut - vCarStrut)
            hLocalToWorld(B[0], rb_carStrut, vCarStrut);
            hLocalToWorld(B[hubB], tyreStrut, vTyreStrut);
            RawJoint j; memset(&j, 0, sizeof(j));
            j.d.type = PDB_JOINT_SLIDER; j.d.b0 = strB; j.d.b1 = hubB; j.d.erp = P.worldErp; j.d.cfm = P.worldCfm; j.d.steerWheel = -1;
            float ax[3]; v3sub(ax, vTyreStrut, vCarStrut); hNorm3(ax);
            hMul1(j.d.axis1, B[strB].R, ax);
            float c[3]; v3sub(c, B[strB].pos, B[hubB].pos);
            hMul1(j.d.offset, B[hubB].R, c);
            hQMul1(j.d.qrel, B[strB].q, B[hubB].q);
            J.push_back(j);
            RawJoint b; memset(&b, 0, sizeof(b));
            b.d.type = PDB_JOINT_BALL; b.d.b0 = PDB_BODY_CHASSIS; b.d.b1 = strB; b.d.erp = P.worldErp; b.d.cfm = P.worldCfm; b.d.steerWheel = -1;
            hWorldToLocal(B[0], vCarStrut, b.d.anchor1);
            hWorldToLocal(B[strB], vCarStrut, b.d.anchor2);
            b.d.qrel[0] = 1;
            J.push_back(b);
            float dl[3]; v3sub(dl, vTyreStrut, vCarStrut);
            S.strutBaseLength = v3len(dl);
        }
        memcpy(S.carStrut, rb_carStrut, sizeof(float) * 3);
        memcpy(S.tyreStrut, tyreStrut, sizeof(float) * 3);
        memcpy(S.tyreSteer, tyreSteer, sizeof(float) * 3);
        memcpy(S.baseCarSteer, rb_carSteer, sizeof(float) * 3);
    }

    // ---- rear rigid axle (SuspensionAxle.cpp:15-118) ----
    for (int index = 2; index < 4; ++index) {
        if (typeR == "DWB") { initDW(index); continue; }
        if (typeR == "ML") { initML(index); continue; }
        pdb_susp& S = P.susp[index];
        memset(&S, 0, sizeof(S));
        S.type = PDB_SUSP_AXLE; S.hubBody = axleB; S.strutBody = -1;
        S.sideSign = (index == 2) ? 1.0f : -1.0f;
        S.axleTrack = rearTrack; S.referenceY = rearBaseY;
        v3set(S.axleBasePos, 0.0f, rearBaseY, -(cg * wheelBase));
        S.attachRelativePos = 1.0f;
        if (iVer >= 4) S.attachRelativePos = sus.getFloat("AXLE", "ATTACH_REL_POS");
        if (index == 2) {
            const float m = sus.getFloat("REAR", "HUB_MASS");
            B[axleB].mass = m; hBoxInertia(m, S.axleTrack * 2.0f, 0.2f, 0.5f, B[axleB].inertia);
            float Mb[9]; hWorldMatrix3(B[0], Mb);
            hSetRotationM(B[axleB], Mb);
            hLocalToWorld(B[0], S.axleBasePos, B[axleB].pos);
            const int links = sus.getInt("AXLE", "LINK_COUNT");
            for (int i = 0; i < links; ++i) {
                char kc[32], ka[32]; snprintf(kc, sizeof(kc), "J%d_CAR", i); snprintf(ka, sizeof(ka), "J%d_AXLE", i);
                float bc[3], ba[3], w[3], relCar[3], relAxle[3], vJ0[3], vJ1[3];
                sus.getFloat3("AXLE", kc, bc); sus.getFloat3("AXLE", ka, ba);
                hLocalToWorld(B[axleB], bc, w); hWorldToLocal(B[0], w, relCar);
                hLocalToWorld(B[axleB], ba, w); hWorldToLocal(B[0], w, relAxle);
                hLocalToWorld(B[0], relCar, vJ0); hLocalToWorld(B[0], relAxle, vJ1);
                mkDBall(J, B, PDB_BODY_CHASSIS, axleB, vJ0, vJ1, P.worldErp, P.worldCfm);
            }
        }
        S.bumpStopUp = sus.getFloat("REAR", "BUMPSTOP_UP");
        S.bumpStopDn = -sus.getFloat("REAR", "BUMPSTOP_DN");
        S.rodLength = sus.getFloat("REAR", "ROD_LENGTH");
        S.toeOutLinear = sus.getFloat("REAR", "TOE_OUT");
        S.k = sus.getFloat("REAR", "SPRING_RATE");
        S.progressiveK = sus.getFloat("REAR", "PROGRESSIVE_SPRING_RATE");
        loadDamper(S.damper, "REAR");
        S.bumpStopRate = sus.getFloat("REAR", "BUMP_STOP_RATE");
        if (S.bumpStopRate == 0.0f) S.bumpStopRate = 500000.0f;
        if (iVer >= 3) S.leafSpringKx = sus.getFloat("AXLE", "LEAF_SPRING_LAT_K");
        v3set(S.basePosition, S.sideSign * S.axleTrack, S.axleBasePos[1], S.axleBasePos[2]);   // getBasePosition()
        S.mass = B[axleB].mass * 0.5f;
    }

    // ---- heave springs (Car.cpp:131-147, HeaveSpring.cpp:11-54): one per axle whose two wheels are double wishbones ----
    for (int a = 0; a < 2; ++a) {
        pdb_heave& H = P.heave[a];
        memset(&H, 0, sizeof(H));
        const std::string id = a ? "HEAVE_REAR" : "HEAVE_FRONT";
        if (P.susp[a * 2].type != PDB_SUSP_DW || P.susp[a * 2 + 1].type != PDB_SUSP_DW || !sus.hasSection(id)) continue;
        H.bumpStopUp = sus.getFloat(id, "BUMPSTOP_UP");
        H.bumpStopDn = -sus.getFloat(id, "BUMPSTOP_DN");
        H.rodLength = sus.getFloat(id, "ROD_LENGTH");
        H.k = sus.getFloat(id, "SPRING_RATE");
        H.progressiveK = sus.getFloat(id, "PROGRESSIVE_SPRING_RATE");
        loadDamper(H.damper, id);
        H.bumpStopRate = sus.getFloat(id, "BUMP_STOP_RATE");
        if (H.bumpStopRate == 0.0f) H.bumpStopRate = 500000.0f;
        H.packerRange = sus.getFloat(id, "PACKER_RANGE");
    }

    // ---- tyres ----
    for (int i = 0; i < 4; ++i) loadTyre(P.tyre[i], P, dataPath, i);
    buildPatchConn(P);
    P.arbK[0] = sus.getFloat("ARB", "FRONT");
    P.arbK[1] = sus.getFloat("ARB", "REAR");
    if (fileExists(dataPath + "ctrl_arb_front.ini")) dynCtrlLoad(P, P.ctrlArb[0], dataPath + "ctrl_arb_front.ini");   // Car.cpp:158-167
    if (fileExists(dataPath + "ctrl_arb_rear.ini")) dynCtrlLoad(P, P.ctrlArb[1], dataPath + "ctrl_arb_rear.ini");
    P.waterTmass = 20.0f; P.waterCoolSpeedK = 0.002f;

    // ---- aero (AeroMap.cpp:15-81, Wing.cpp:19-69) ----
    {
        Ini aero(dataPath + "aero.ini");
        if (!aero.ready) throw std::runtime_error("pdb: aero.ini not found");
        // [SLIPSTREAM] (AeroMap.cpp:25-29; no shipped car carries it: SlipStream.h's defaults then)
        P.slipEffectGainMult = 1.0f; P.slipSpeedFactorMult = 1.0f;
        if (aero.hasSection("SLIPSTREAM")) { P.slipEffectGainMult = aero.getFloat("SLIPSTREAM", "EFFECT_GAIN_MULT"); P.slipSpeedFactorMult = aero.getFloat("SLIPSTREAM", "SPEED_FACTOR_MULT"); }
        const int aver = aero.getInt("HEADER", "VERSION");
        int n = 0;
        for (int pass = 0; pass < 2; ++pass)
            for (int id = 0;; ++id) {
                char secn[32]; snprintf(secn, sizeof(secn), "%s_%d", pass ? "FIN" : "WING", id);
                if (!aero.hasSection(secn)) break;
                if (n >= PDB_MAX_WINGS) throw std::runtime_error("pdb: too many wings");
                pdb_wing& W = P.wings[n++];
                memset(&W, 0, sizeof(W));
                W.isVertical = pass;
                const float chord = aero.getFloat(secn, "CHORD"), span = aero.getFloat(secn, "SPAN");
                W.area = chord * span;
                aero.getFloat3(secn, "POSITION", W.position);
                curveLoad(W.lutAOA_CL, dataPath + aero.getString(secn, "LUT_AOA_CL"));
                curveLoad(W.lutAOA_CD, dataPath + aero.getString(secn, "LUT_AOA_CD"));
                const std::string ghcl = aero.getString(secn, "LUT_GH_CL"), ghcd = aero.getString(secn, "LUT_GH_CD");
                if (!ghcl.empty() && fileExists(dataPath + ghcl)) curveLoad(W.lutGH_CL, dataPath + ghcl);   // Wing.cpp:38-44: loaded when the file is there
                if (!ghcd.empty() && fileExists(dataPath + ghcd)) curveLoad(W.lutGH_CD, dataPath + ghcd);
                if (W.lutGH_CL.n > 0 || W.lutGH_CD.n > 0) P.wingGroundEffect = 1;
                W.cdGain = aero.getFloat(secn, "CD_GAIN");
                W.clGain = aero.getFloat(secn, "CL_GAIN");
                W.angle = aero.getFloat(secn, "ANGLE");
                if (aver >= 3) W.yawGain = aero.getFloat(secn, "YAW_CL_GAIN");
            }
        P.aeroReferenceArea = 1.0f; P.aeroFrontShare = 0.5f; P.aeroCD = 0.0f; P.aeroCL = 0.0f; P.aeroCDX = 0.0f; P.aeroCDY = 0.0f; P.aeroCDA = 0.1f;   // AeroMap.h:20-26
        if (n == 0) {   // AeroMap.cpp:49-58: no wings at all -> the map's own coefficients
            if (!aero.hasSection("DATA")) throw std::runtime_error("pdb: aero.ini has neither [WING_n] / [FIN_n] nor [DATA]");
            P.aeroReferenceArea = aero.getFloat("DATA", "REFERENCE_AREA"); P.aeroFrontShare = aero.getFloat("DATA", "FRONT_SHARE");
            P.aeroCD = aero.getFloat("DATA", "CD"); P.aeroCL = aero.getFloat("DATA", "CL"); P.aeroCDX = aero.getFloat("DATA", "CDX"); P.aeroCDY = aero.getFloat("DATA", "CDY");
        }
        // [DYNAMIC_CONTROLLER_n] (AeroMap.cpp:66-82; WingDynamicController::init): a filtered LUT of a car signal added to / multiplied into a wing's angle
        P.numWingCtrl = 0;
        for (int id = 0;; ++id) {
            char secn[40]; snprintf(secn, sizeof(secn), "DYNAMIC_CONTROLLER_%d", id);
            if (!aero.hasSection(secn)) break;
            const int iWing = aero.getInt(secn, "WING");
            if (iWing < 0 || iWing >= n) continue;   // (the reference warns and goes on)
            if (P.numWingCtrl >= PDB_MAX_WING_CTRL) throw std::runtime_error("pdb: more than " + std::to_string(PDB_MAX_WING_CTRL) + " wing dynamic controllers");
            pdb_wing_ctrl& wc = P.wingCtrl[P.numWingCtrl++];
            memset(&wc, 0, sizeof(wc));
            wc.wing = iWing;
            static const char* inputs[] = {"", "BRAKE", "GAS", "LATG", "LONG", "STEER", "SPEED_KMH", "SUS_TRAVEL_LR", "SUS_TRAVEL_RR"};
            const std::string in = aero.getString(secn, "INPUT"), comb = aero.getString(secn, "COMBINATOR");
            for (int k = 1; k <= 8; ++k) if (in == inputs[k]) wc.input = k;
            wc.combinator = comb == "ADD" ? 1 : comb == "MULT" ? 2 : 0;
            if (wc.input == 0 || wc.combinator == 0) throw std::runtime_error(std::string("pdb: ") + secn + ": unknown INPUT / COMBINATOR (the reference stops at its first step)");
            if (wc.input >= 7) P.wingGroundEffect = 1;   // reads this tick's suspension travel: the wings then step after the force barrier, like those with ground-effect LUTs
            curveLoad(wc.lut, dataPath + aero.getString(secn, "LUT"));
            wc.filter = ((1.0f - aero.getFloat(secn, "FILTER")) * 1.3333334f) * 333.33334f;
            wc.upLimit = aero.getFloat(secn, "UP_LIMIT");
            wc.downLimit = aero.getFloat(secn, "DOWN_LIMIT");
        }
        P.numWings = n;
    }

    // ---- brakes (BrakeSystem.cpp:14-73) ----
    {
        Ini br(dataPath + "brakes.ini");
        if (!br.ready) throw std::runtime_error("pdb: brakes.ini not found");
        P.brakePower = br.getFloat("DATA", "MAX_TORQUE");
        P.frontBias = br.getFloat("DATA", "FRONT_SHARE");
        P.handBrakeTorque = br.getFloat("DATA", "HANDBRAKE_TORQUE");
        P.brakePowerMultiplier = 1.0f;
        P.biasMin = 0.0f; P.biasMax = 1.0f;
        if (br.hasSection("TEMPS_FRONT") && br.hasSection("TEMPS_REAR")) {   // BrakeSystem.cpp:40-52
            P.hasBrakeTemps = 1;
            for (int id = 0; id < 4; ++id) {
                const char* sec = id < 2 ? "TEMPS_FRONT" : "TEMPS_REAR";
                pdb_brake_disc& d = P.discs[id];
                const std::string v = br.getString(sec, "PERF_CURVE");
                if (v.find(".lut") != std::string::npos) curveLoad(d.perfCurve, dataPath + v); else curveParseInline(d.perfCurve, v);
                d.torqueK = br.getFloat(sec, "TORQUE_K"); d.coolTransfer = br.getFloat(sec, "COOL_TRANSFER"); d.coolSpeedFactor = br.getFloat(sec, "COOL_SPEED_FACTOR");
            }
        }
        if (fileExists(dataPath + "steer_brake_controller.ini")) {   // BrakeSystem.cpp:33-38
            dynCtrlLoad(P, P.ctrlSteerBrake, dataPath + "steer_brake_controller.ini");
            if (P.ctrlSteerBrake.count == 0) throw std::runtime_error("pdb: steer_brake_controller.ini has no usable stage");
        }
        if (br.hasSection("EBB")) {   // EBBMode::Internal (BrakeSystem.cpp:28-32): the bias follows the front axle's share of the load
            P.ebbInternal = 1;
            const float m = br.getFloat("EBB", "FRONT_SHARE_MULTIPLIER");
            P.ebbFrontMultiplier = m > 1.1f ? m : 1.1f;
        }
        if (fileExists(dataPath + "ctrl_ebb.ini")) {   // EBBMode::DynamicController (BrakeSystem.cpp:64-69): takes the place of the internal mode
            dynCtrlLoad(P, P.ctrlEbb, dataPath + "ctrl_ebb.ini");
            if (P.ctrlEbb.count == 0) throw std::runtime_error("pdb: ctrl_ebb.ini has no usable stage");
            P.ebbInternal = 0;
        }
        Ini setup(dataPath + "setup.ini");
        if (setup.ready && setup.hasSection("FRONT_BIAS")) {
            P.biasMin = setup.getFloat("FRONT_BIAS", "MIN") * 0.01f;
            P.biasMax = setup.getFloat("FRONT_BIAS", "MAX") * 0.01f;
        }
    }

    // ---- engine (Engine.cpp:16-191) ----
    Ini eng(dataPath + "engine.ini");
    if (!eng.ready) throw std::runtime_error("pdb: engine.ini not found");
    {
        curveLoad(P.powerCurve, dataPath + eng.getString("HEADER", "POWER_CURVE"));
        P.engMinimum = eng.getInt("ENGINE_DATA", "MINIMUM");
        if (!P.engMinimum) P.engMinimum = 1000;
        P.engCoast1 = 0.0f; P.engCoast2 = 0.000001f;   // EngineData defaults (Engine.h:18-20)
        if (eng.getString("HEADER", "COAST_CURVE") == "FROM_COAST_REF") {
            const float rpm = eng.getFloat("COAST_REF", "RPM"), tq = eng.getFloat("COAST_REF", "TORQUE"), nl = eng.getFloat("COAST_REF", "NON_LINEARITY");
            const float v13 = ((1.0f - nl) * rpm) - P.engMinimum;
            const float v14 = nl * rpm;
            P.engCoast1 = (v13 == 0.0f) ? 0.0f : -(tq / v13);
            P.engCoast2 = (v14 == 0.0f) ? 0.0f : tq / (v14 * v14);
        }
        P.engInertia = eng.getFloat("ENGINE_DATA", "INERTIA");
        P.engLimiter = eng.getInt("ENGINE_DATA", "LIMITER");
        if (P.engLimiter) { P.rpmDamageThreshold = P.engLimiter * 1.05f; P.rpmDamageK = 10.0f; }
        int hz = eng.getInt("ENGINE_DATA", "LIMITER_HZ");
        P.engLimiterCycles = hz ? (1000 / hz / 3) : 50;
        if (eng.hasSection("COAST_SETTINGS")) {   // Engine.cpp:61-67, setCoastSettings :394-399: the offset is the LUT's value at the DEFAULT index
            pdb_curve off; memset(&off, 0, sizeof(off));
            const std::string lutv = eng.getString("COAST_SETTINGS", "LUT");
            if (lutv.find(".lut") != std::string::npos) curveLoad(off, dataPath + lutv); else curveParseInline(off, lutv);
            const int id = eng.getInt("COAST_SETTINGS", "DEFAULT");
            if (id >= 0 && id < off.n) P.gasCoastOffset = curveValue(off, (float)id);
            P.coastEntryRpm = P.engMinimum + eng.getInt("COAST_SETTINGS", "ACTIVATION_RPM");
        }
        // turbos (Engine.cpp:69-94): TURBO_0.. until a section is missing; cockpit-adjustable ones take the default adjustment
        bool adjustable = false;
        for (int id = 0; ; ++id) {
            char sec[32]; snprintf(sec, sizeof(sec), "TURBO_%d", id);
            if (!eng.hasSection(sec)) break;
            if (id >= PDB_MAX_TURBOS) throw std::runtime_error("pdb: more than 3 turbos");
            pdb_turbo& tb = P.turbos[id];
            tb.lagDN = (1.0f - eng.getFloat(sec, "LAG_DN")) * 1.333333f * 333.3333f;
            tb.lagUP = (1.0f - eng.getFloat(sec, "LAG_UP")) * 1.333333f * 333.3333f;
            tb.maxBoost = eng.getFloat(sec, "MAX_BOOST");
            tb.wastegate = eng.getFloat(sec, "WASTEGATE");
            tb.rpmRef = eng.getFloat(sec, "REFERENCE_RPM");
            tb.gamma = eng.getFloat(sec, "GAMMA");
            tb.isAdjustable = (eng.getInt(sec, "COCKPIT_ADJUSTABLE") != 0) ? 1 : 0;
            tb.userSetting = 1.0f;   // Turbo.h default
            if (tb.isAdjustable) adjustable = true;
            P.numTurbos = id + 1;
        }
        if (adjustable) {   // setTurboBoostLevel (Turbo.cpp:46-52)
            const float boost = eng.getFloat("ENGINE_DATA", "DEFAULT_TURBO_ADJUSTMENT");
            for (int id = 0; id < P.numTurbos; ++id) P.turbos[id].userSetting = P.turbos[id].isAdjustable ? boost : 1.0f;
        }
        if (P.numTurbos > 0) {
            for (int id = 0; id < P.numTurbos; ++id) {
                char c1[64], c2[64]; snprintf(c1, sizeof(c1), "ctrl_turbo%d.ini", id); snprintf(c2, sizeof(c2), "ctrl_wastegate%d.ini", id);
                if (fileExists(dataPath + c1)) dynCtrlLoad(P, P.ctrlTurboBoost[id], dataPath + c1);   // Engine.cpp:124-143
                if (fileExists(dataPath + c2)) dynCtrlLoad(P, P.ctrlWastegate[id], dataPath + c2);
            }
        }
        if (eng.hasSection("OVERLAP")) {   // Engine.cpp:96-101
            P.overlapFreq = eng.getFloat("OVERLAP", "FREQUENCY"); P.overlapGain = eng.getFloat("OVERLAP", "GAIN"); P.overlapIdealRPM = eng.getFloat("OVERLAP", "IDEAL_RPM");
        }
        curveLoad(P.throttleCurve, dataPath + "throttle.lut");
        if (eng.hasSection("THROTTLE_RESPONSE")) {   // Engine.cpp:150-154: a second curve, blended in by rpm / RPM_REFERENCE (getThrottleResponseGas :344-366)
            P.throttleMaxRef = eng.getFloat("THROTTLE_RESPONSE", "RPM_REFERENCE");
            const std::string lutv = eng.getString("THROTTLE_RESPONSE", "LUT");
            if (lutv.find(".lut") != std::string::npos) curveLoad(P.throttleCurveMax, dataPath + lutv); else curveParseInline(P.throttleCurveMax, lutv);
        }
        if (eng.hasSection("DAMAGE")) {
            P.rpmDamageThreshold = eng.getFloat("DAMAGE", "RPM_THRESHOLD");
            P.rpmDamageK = eng.getFloat("DAMAGE", "RPM_DAMAGE_K");
            if (P.numTurbos > 0) {
                P.turboBoostDamageThreshold = eng.getFloat("DAMAGE", "TURBO_BOOST_THRESHOLD");
                P.turboBoostDamageK = eng.getFloat("DAMAGE", "TURBO_DAMAGE_K");
            }
        }
        P.bovThreshold = 0.2f;
        if (eng.hasSection("BOV")) P.bovThreshold = eng.getFloat("BOV", "PRESSURE_THRESHOLD");
        P.limiterMultiplier = 1.0f;
        // precalculatePowerAndTorque (Engine.cpp:170-191)
        const float maxRef = P.powerCurve.n ? P.powerCurve.x[P.powerCurve.n - 1] : 0.0f;
        float maxTq = 0, maxPw = 0;
        for (float rpm = 0; rpm <= maxRef; rpm += 50.0f) {
            const float tq = curveValue(P.powerCurve, rpm);
            if (tq > maxTq) { P.maxTorqueRPM = rpm; maxTq = tq; }
            const float pw = rpm * tq * 0.1047f;
            if (pw > maxPw) { P.maxPowerRPM = rpm; maxPw = pw; }
        }
    }

    // ---- drivetrain (Drivetrain.cpp:18-152) ----
    {
        Ini dt(dataPath + "drivetrain.ini");
        if (!dt.ready) throw std::runtime_error("pdb: drivetrain.ini not found");
        const std::string tr = dt.getString("TRACTION", "TYPE");
        if (tr != "RWD" && tr != "FWD") throw std::runtime_error("pdb: traction type " + tr + " unsupported (reference: TODO_NOT_IMPLEMENTED_FATAL)");
        P.tractionType = (tr == "RWD") ? 0 : 1;
        for (int i = 0; i < 4; ++i) P.tyre[i].driven = (P.tractionType == 0) ? (i >= 2) : (i < 2);
        P.damageRpmWindow = dt.getFloat("DAMAGE", "RPM_WINDOW_K");
        int ng = 0;
        P.gearRatio[ng++] = dt.getFloat("GEARS", "GEAR_R");
        P.gearRatio[ng++] = 0.0f;
        const int cnt = dt.getInt("GEARS", "COUNT");
        for (int i = 1; i <= cnt; ++i) {
            char k[16]; snprintf(k, sizeof(k), "GEAR_%d", i);
            if (ng >= PDB_MAX_GEARS) throw std::runtime_error("pdb: too many gears");
            P.gearRatio[ng++] = dt.getFloat("GEARS", k);
        }
        P.numGears = ng;
        P.finalRatio = dt.getFloat("GEARS", "FINAL");
        P.diffPowerRamp = dt.getFloat("DIFFERENTIAL", "POWER");
        P.diffCoastRamp = dt.getFloat("DIFFERENTIAL", "COAST");
        P.diffPreLoad = dt.getFloat("DIFFERENTIAL", "PRELOAD");
        P.diffType = (P.diffPowerRamp >= 1.0f && P.diffCoastRamp >= 1.0f) ? 1 : 0;
        P.gearUpTime = dt.getFloat("GEARBOX", "CHANGE_UP_TIME") * 0.001f;
        if (P.gearUpTime == 0.0f) P.gearUpTime = 0.1f;
        P.gearDnTime = dt.getFloat("GEARBOX", "CHANGE_DN_TIME") * 0.001f;
        if (P.gearDnTime == 0.0f) P.gearDnTime = 0.15f;
        P.autoCutOffTime = dt.getFloat("GEARBOX", "AUTO_CUTOFF_TIME") * 0.001f;
        P.isShifterSupported = dt.getInt("GEARBOX", "SUPPORTS_SHIFTER") != 0;
        P.validShiftRPMWindow = dt.getFloat("GEARBOX", "VALID_SHIFT_RPM_WINDOW");
        if (P.validShiftRPMWindow == 0.0) P.validShiftRPMWindow = 500.0;
        P.controlsWindowGain = dt.getFloat("GEARBOX", "CONTROLS_WINDOW_GAIN");
        P.engineInertiaInit = 0.01f; P.driveInertia = 0.01f; P.clutchInertia = 1.0;
        const float gi = dt.getFloat("GEARBOX", "INERTIA");
        if (gi != 0.0f) { P.clutchInertia = gi; P.driveInertia = gi; }
        P.clutchMaxTorque = dt.getFloat("CLUTCH", "MAX_TORQUE");
        if (P.clutchMaxTorque == 0.0) P.clutchMaxTorque = 450.0;
        const int tl = (P.tractionType == 0) ? 2 : 0;
        P.outShaftInertiaL = P.tyre[tl].angularInertia;
        P.outShaftInertiaR = P.tyre[tl + 1].angularInertia;
        if (P.tractionType == 0 && fileExists(dataPath + "ctrl_single_lock.ini")) dynCtrlLoad(P, P.ctrlDiffLock, dataPath + "ctrl_single_lock.ini");   // Drivetrain.cpp:144-151
        // AutoClutch (AutoClutch.cpp:25-89)
        const std::string up = dt.getString("AUTOCLUTCH", "UPSHIFT_PROFILE"), dn = dt.getString("AUTOCLUTCH", "DOWNSHIFT_PROFILE");
        P.acUseOnChange = dt.getInt("AUTOCLUTCH", "USE_ON_CHANGES") != 0;
        auto prof = [&](pdb_curve& c, const std::string& name) {
            c.n = 0;
            if (name != "NONE" && dt.hasSection(name)) {
                curveAdd(c, 0.0f, 1.0f);
                curveAdd(c, dt.getFloat(name, "POINT_0") * 0.001f, 0.0f);
                curveAdd(c, dt.getFloat(name, "POINT_1") * 0.001f, 0.0f);
                curveAdd(c, dt.getFloat(name, "POINT_2") * 0.001f, 1.0f);
            }
        };
        prof(P.upshiftProfile, up); prof(P.downshiftProfile, dn);
        P.acRpmMin = dt.getFloat("AUTOCLUTCH", "MIN_RPM"); P.acRpmMax = dt.getFloat("AUTOCLUTCH", "MAX_RPM");
        if (P.acRpmMin == 0.0f || P.acRpmMax == 0.0f) { P.acRpmMin = 1500.0f; P.acRpmMax = 2500.0f; }
        P.acClutchSpeed = 1.0f;
        // AutoBlip (AutoBlip.cpp:14-49)
        const float lvl = dt.getFloat("AUTOBLIP", "LEVEL");
        P.blipProfile.n = 0;
        curveAdd(P.blipProfile, 0.0f, 0.0f);
        curveAdd(P.blipProfile, dt.getFloat("AUTOBLIP", "POINT_0"), lvl);
        curveAdd(P.blipProfile, dt.getFloat("AUTOBLIP", "POINT_1"), lvl);
        curveAdd(P.blipProfile, dt.getFloat("AUTOBLIP", "POINT_2"), 0.0f);
        P.blipPerformTime = P.blipProfile.x[3];
        P.autoBlipElectronic = dt.getInt("AUTOBLIP", "ELECTRONIC") != 0;
        // AutoShifter (AutoShifter.cpp:14-29, defaults AutoShifter.h:16-20)
        P.asChangeUpRpm = 0; P.asChangeDnRpm = 4000; P.asSlipThreshold = 0.8f; P.asGasCutoffTime = 0.5f;
        if (dt.hasSection("AUTO_SHIFTER")) {
            P.asChangeUpRpm = dt.getInt("AUTO_SHIFTER", "UP");
            P.asChangeDnRpm = dt.getInt("AUTO_SHIFTER", "DOWN");
            P.asSlipThreshold = dt.getFloat("AUTO_SHIFTER", "SLIP_THRESHOLD");
            P.asGasCutoffTime = dt.getFloat("AUTO_SHIFTER", "GAS_CUTOFF_TIME");
        }
        if (!P.asChangeUpRpm) {   // lazy init in AutoShifter::step (AutoShifter.cpp:38-54)
            const float lim = (float)(int)(P.engLimiter * P.limiterMultiplier);
            const float mx = (lim >= P.maxPowerRPM) ? P.maxPowerRPM : lim;
            P.asChangeUpRpm = (int)(mx * 0.98f);
            P.asChangeDnRpm = (int)(P.maxTorqueRPM * 1.1f);
        }
    }
    P.acUseOnStart = 0; P.autoShiftActive = 0; P.autoBlipActive = 0; P.smoothSteer = 0;

    // ---- body masses (Car::updateBodyMass / calcBodyMass, Car.cpp:589-620) ----
    {
        float suspMass = 0;
        for (int i = 0; i < 4; ++i) suspMass += P.susp[i].mass;
        const float bodyMass = (P.mass - suspMass) + 0.0f;
        B[0].mass = bodyMass; hBoxInertia(bodyMass, bodyInertia[0], bodyInertia[1], bodyInertia[2], B[0].inertia);
        const float fuelMass = std::max(0.1f, P.fuelKG * (float)(double)P.fuel);
        B[1].mass = fuelMass; hBoxInertia(fuelMass, 0.5f, 0.5f, 0.5f, B[1].inertia);
    }
    P.numBodies = nextBody;
    for (int i = 0; i < P.numBodies; ++i) { P.bodies[i].mass = B[i].mass; memcpy(P.bodies[i].inertia, B[i].inertia, sizeof(float) * 3); }

    // ---- joints into solver order ----
    if ((int)J.size() > PDB_MAX_JOINTS) throw std::runtime_error("pdb: too many joints");
    const std::vector<int> order = islandOrder(J, P.numBodies);
    P.numJoints = (int)order.size();
    int rows = 0;
    for (int t = 0; t < P.numJoints; ++t) {
        P.joints[t] = J[order[t]].d;
        const int ty = P.joints[t].type;
        rows += (ty == PDB_JOINT_FIXED) ? 6 : (ty == PDB_JOINT_BALL) ? 3 : (ty == PDB_JOINT_SLIDER) ? 5 : 1;
    }
    if (rows > PDB_MAX_ROWS) throw std::runtime_error("pdb: too many constraint rows");
    P.numRows = rows;

    // getBaseCarHeight (Car.cpp:1360-1365)
    {
        const float t0 = fabsf(P.susp[0].basePosition[1] - P.tyre[0].rimRadius);
        const float t2 = fabsf(P.susp[2].basePosition[1] - P.tyre[2].rimRadius);
        P.baseCarHeight = std::max(t0, t2);
    }
    // the device evaluates LUTs by counting knots below the argument: every curve must be sorted
    {
        auto sorted = [](const pdb_curve& c, const char* what) {
            for (int i = 1; i < c.n; ++i) if (c.x[i] < c.x[i - 1]) throw std::runtime_error(std::string("pdb: LUT with decreasing abscissae: ") + what);
        };
        sorted(P.powerCurve, "power"); sorted(P.throttleCurve, "throttle"); sorted(P.throttleCurveMax, "throttle response"); sorted(P.upshiftProfile, "upshift"); sorted(P.downshiftProfile, "downshift");
        sorted(P.blipProfile, "blip");
        for (int i = 0; i < 4; ++i) { sorted(P.tyre[i].performanceCurve, "tyre performance"); sorted(P.tyre[i].wearCurve, "tyre wear"); }
        for (int i = 0; i < P.numWings; ++i) { sorted(P.wings[i].lutAOA_CL, "wing CL"); sorted(P.wings[i].lutAOA_CD, "wing CD"); }
    }
    // ---- body colliders: CarColliderManager::init (CarColliderManager.cpp:12-37), Car::loadColliderBlob / initColliderMesh
    //      (Car.cpp:318-377).  Both files are optional here (the reference insists on them); without them the car has no
    //      body contacts.  The hull's vertices go through the geom offset addMeshCollider installs (RigidBodyODE.cpp:294-317):
    //      getGraphicsOffsetMatrix() of the freshly created body = pitch rotation about x, then the graphics offset. ----
    {
        pdb_collider& C = P.collider;
        memset(&C, 0, sizeof(C));
        float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        auto grow = [&](const float* v) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], v[k]); hi[k] = std::max(hi[k], v[k]); } };
        Ini col(dataPath + "colliders.ini");
        for (int id = 0; col.ready; ++id) {   // CarColliderManager.cpp:17-33: every COLLIDER_n there is, each a box geom of its own on the chassis
            char sec[32]; snprintf(sec, sizeof(sec), "COLLIDER_%d", id);
            if (!col.hasSection(sec)) break;
            if (id >= PDB_MAX_BOXES) throw std::runtime_error("pdb: more than " + std::to_string(PDB_MAX_BOXES) + " box colliders unsupported");
            float size[3];
            col.getFloat3(sec, "CENTRE", C.boxCentre[id]);
            col.getFloat3(sec, "SIZE", size);
            for (int k = 0; k < 3; ++k) C.boxHalf[id][k] = size[k] * 0.5f;
            C.numBoxes = id + 1;
            for (int c = 0; c < 8; ++c) {
                const float v[3] = {C.boxCentre[id][0] + ((c & 1) ? C.boxHalf[id][0] : -C.boxHalf[id][0]), C.boxCentre[id][1] + ((c & 2) ? C.boxHalf[id][1] : -C.boxHalf[id][1]),
                                    C.boxCentre[id][2] + ((c & 4) ? C.boxHalf[id][2] : -C.boxHalf[id][2])};
                grow(v);
            }
        }
        if (FILE* f = fopen((dataPath + "collider.bin").c_str(), "rb")) {
            uint32_t hdr[3] = {0, 0, 0};   // magic, numVertices, numIndices (Car.cpp:320-327)
            const bool okH = fread(hdr, sizeof(hdr), 1, f) == 1;
            std::vector<float> vb; std::vector<uint16_t> ib;
            bool ok = okH && hdr[1] > 0 && hdr[2] > 0 && hdr[1] <= 65536 && hdr[2] <= 1000000;
            if (ok) { vb.resize((size_t)hdr[1] * 3); ib.resize(hdr[2]); ok = fread(vb.data(), vb.size() * 4, 1, f) == 1 && fread(ib.data(), ib.size() * 2, 1, f) == 1; }
            fclose(f);
            if (!ok) throw std::runtime_error("pdb: malformed collider.bin for " + modelName);
            if (hdr[1] > PDB_MAX_COLL_VERTS || hdr[2] / 3 > PDB_MAX_COLL_TRIS) throw std::runtime_error("pdb: collider.bin too large (max 128 vertices, 192 triangles)");
            float off[3] = {0, 0, 0};
            if (car.hasKey("BASIC", "GRAPHICS_OFFSET")) car.getFloat3("BASIC", "GRAPHICS_OFFSET", off);
            const float pitch = (float)((double)(car.hasKey("BASIC", "GRAPHICS_PITCH_ROTATION") ? car.getFloat("BASIC", "GRAPHICS_PITCH_ROTATION") : 0.0f) * (3.14159265358979323846 / (double)180.0f));
            float M[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            const float ax[3] = {1, 0, 0};
            if (pitch != 0.0f) hAxisAngle(ax, pitch, M);
            C.numVerts = (int32_t)hdr[1]; C.numTris = (int32_t)(hdr[2] / 3);
            for (int i = 0; i < C.numVerts; ++i) {
                const float* v = vb.data() + 3 * (size_t)i;
                for (int c = 0; c < 3; ++c) C.verts[i][c] = off[c] + (v[0] * M[c] + v[1] * M[3 + c] + v[2] * M[6 + c]);   // v * gm (row vector)
                grow(C.verts[i]);
            }
            for (int t = 0; t < C.numTris; ++t)
                for (int k = 0; k < 3; ++k) {
                    if (ib[3 * (size_t)t + k] >= hdr[1]) throw std::runtime_error("pdb: collider.bin index out of range");
                    C.tris[t][k] = (uint8_t)ib[3 * (size_t)t + k];
                }
        }
        if (C.numBoxes || C.numTris) { for (int k = 0; k < 3; ++k) { C.boundsLo[k] = lo[k]; C.boundsHi[k] = hi[k]; } C.enabled = 1; }
    }
    // ScoringConfig defaults (ScoringSystem.cpp:46-71)
    pdb_scoring& sc = P.scoring;
    memset(&sc, 0, sizeof(sc));
    sc.SmoothSteerSpeed = 10.0f; sc.MinBonusSpeed = 5.0f; sc.MaxBonusSpeed = 200.0f; sc.StallRpm = 300.0f;
    sc.DirectionThreshold = 0.75f; sc.OutOfTrackThreshold = 0.51f; sc.ApproachDistance = 3.0f; sc.CriticalDistance = 2.0f;
}

// -------------------------------------------------------------------------------------------------
// tunes (Car/SetupManager.cpp:10-120,355-444) -- the subset of SetupVar needed for float/double
// variables; spinner semantics from setup.ini (SHOW_CLICKS, MIN, MAX, STEP) and final.rto.
// -------------------------------------------------------------------------------------------------
namespace {
struct TuneVar { float* f; double* d; float mult; bool preset; /* tunable with a built-in range, setup.ini or not */ };
static bool findTune(pdb_car_params& P, const std::string& name, TuneVar& v) {
    v.f = nullptr; v.d = nullptr; v.mult = 1.0f; v.preset = false;
    static const char* types[] = {"LF", "RF", "LR", "RR"};
    static const double sides[] = {-1, 1, -1, 1};
    if (name == "FRONT_BIAS") { v.f = &P.frontBias; v.mult = (float)0.01; return true; }
    if (name == "BRAKE_POWER_MULT") { v.f = &P.brakePowerMultiplier; v.mult = (float)0.01; return true; }
    if (name == "DIFF_POWER") { v.d = &P.diffPowerRamp; v.mult = (float)0.01; return true; }
    if (name == "DIFF_COAST") { v.d = &P.diffCoastRamp; v.mult = (float)0.01; return true; }
    if (name == "DIFF_PRELOAD") { v.d = &P.diffPreLoad; return true; }
    if (name == "FINAL_RATIO") { v.d = &P.finalRatio; return true; }
    if (name == "ARB_FRONT") { v.f = &P.arbK[0]; return true; }
    if (name == "ARB_REAR") { v.f = &P.arbK[1]; return true; }
    if (name == "ENGINE_LIMITER") { v.f = &P.limiterMultiplier; v.mult = (float)0.01; return true; }
    for (int g = 0; g < P.numGears; ++g) {
        char b[32]; snprintf(b, sizeof(b), "INTERNAL_GEAR_%d", g);
        if (name == b) { v.d = &P.gearRatio[g]; return true; }
    }
    for (int t = 0; t < P.numTurbos; ++t) {   // SetupManager.cpp:93-106: userSetting, tunable on its own (0..1 in steps of 0.1), no setup.ini needed
        char b[32]; snprintf(b, sizeof(b), "TURBO_%d", t);
        if (name == b) { v.f = &P.turbos[t].userSetting; v.mult = (float)0.01; v.preset = true; return true; }
    }
    for (int w = 0; w < P.numWings; ++w) {
        // WING_n of a wing without a dynamic controller (the only kind the loader accepts) is bound to Wing::status.angle
        // (SetupManager.cpp:112-125), which Wing::stepDynamicControllers overwrites with status.inputAngle -- the aero.ini
        // angle -- on every tick (Wing.cpp:73-74,105-123): the tune is accepted and has no effect.  Pinned by the `tunes` goldens.
        static float overwrittenEveryTick;
        char b[32]; snprintf(b, sizeof(b), "WING_%d", w);
        if (name == b) { v.f = &overwrittenEveryTick; return true; }
    }
    for (int i = 0; i < 4; ++i) {
        const std::string t = types[i];
        pdb_susp& S = P.susp[i];
        if (name == "DAMP_FAST_BUMP_" + t) { v.f = &S.damper.bumpFast; return true; }
        if (name == "DAMP_BUMP_" + t) { v.f = &S.damper.bumpSlow; return true; }
        if (name == "DAMP_FAST_REBOUND_" + t) { v.f = &S.damper.reboundFast; return true; }
        if (name == "DAMP_REBOUND_" + t) { v.f = &S.damper.reboundSlow; return true; }
        if (name == "BUMP_STOP_RATE_" + t) { v.f = &S.bumpStopRate; v.mult = (float)1000.0; return true; }
        if (name == "SPRING_RATE_" + t) { v.f = &S.k; v.mult = (float)1000.0; return true; }
        if (name == "PROGRESSIVE_SPRING_RATE_" + t) { v.f = &S.progressiveK; v.mult = (float)1000.0; return true; }
        if (name == "ROD_LENGTH_" + t) { v.f = &S.rodLength; v.mult = (float)0.0001; return true; }
        if (name == "CAMBER_" + t) { v.f = &S.staticCamber; v.mult = (float)(0.0017453292 * sides[i]); return true; }
        if (name == "TOE_OUT_" + t) { v.f = &S.toeOutLinear; v.mult = (float)0.00001; return true; }
        if (name == "PACKER_RANGE_" + t) { v.f = &S.packerRange; v.mult = (float)0.001; return true; }
        if (name == "PRESSURE_" + t) { v.f = &P.tyre[i].pressureStatic; return true; }
    }
    return false;
}
static inline float truncF(float x) { return (float)(int)x; }
}  // namespace

bool setCarTune(pdb_car_params& P, const std::string& basePathIn, const std::string& modelName, const std::string& name, float value, bool raw) {
    TuneVar v;
    if (!findTune(P, name, v)) return false;   // PyProjectD.cpp:328-345: unknown names are ignored
    auto setRaw = [&](float x) { if (v.f) *v.f = x; else *v.d = x; };
    if (raw) { setRaw(value); return true; }
    std::string base = basePathIn;
    std::replace(base.begin(), base.end(), '\\', '/');
    if (!base.empty() && base.back() != '/') base += '/';
    const std::string dataPath = base + "content/cars/" + modelName + "/data/";
    // SetupVar defaults (SetupManager.h:29-44)
    float minV = -3.402823466e+38f, maxV = 3.402823466e+38f, step = 0.01f;
    int spinner = 3;  // RawFloat
    bool tunable = false;
    if (v.preset) { minV = 0.0f; maxV = 1.0f; step = 0.1f; tunable = true; }
    std::vector<float> predefined;
    Ini ini(dataPath + "setup.ini");
    if (!ini.ready) {
        // a car loaded from a packed block has no data/ directory: its tunes are already applied in the block, and
        // the spinner ranges that turn a setup value into a parameter are not available
        FILE* probe = fopen((dataPath + "car.ini").c_str(), "rb");
        if (probe) fclose(probe);
        else throw std::runtime_error("pdb: setCarTune(" + name + ") needs " + dataPath + "setup.ini (packed car blocks carry their tunes pre-applied)");
    }
    if (ini.ready) {
        if (name == "FINAL_RATIO" && ini.hasKey("FINAL_GEAR_RATIO", "RATIOS")) {
            std::ifstream fs(dataPath + ini.getString("FINAL_GEAR_RATIO", "RATIOS"));
            std::string line;
            while (fs.is_open() && std::getline(fs, line)) {
                if (!line.empty() && line.back() == '\r') line.pop_back();
                if (line.empty()) continue;
                auto kv = splitStr(line, "|");
                if (kv.size() == 2) predefined.push_back(toFloat(kv[1]));
            }
            if (!predefined.empty()) {
                std::sort(predefined.begin(), predefined.end());
                spinner = 4;  // RawPredefined
                minV = predefined.front(); maxV = predefined.back(); step = 0.01f; tunable = true;
            }
        }
        if (ini.hasSection(name)) {
            const bool t = ini.tryGetFloat(name, "MIN", minV) && ini.tryGetFloat(name, "MAX", maxV) && ini.tryGetFloat(name, "STEP", step);
            int sc = 0;
            if (ini.tryGetInt(name, "SHOW_CLICKS", sc) && sc >= 0 && sc <= 4) spinner = sc;
            if (t) tunable = true;
        }
    }
    if (tunable && (minV >= maxV || fabs(maxV - minV) < 0.01)) { minV = 0; maxV = 0; tunable = false; }
    if (tunable && step < 0.0f) { step = 0.0f; tunable = false; }
    if (!tunable) spinner = 3;
    // getSpinner(type) limits, then setValue (SetupManager.cpp:355-421)
    float smin, smax;
    switch (spinner) {
        case 0: smin = truncF(minV); smax = truncF(maxV); break;
        case 1: smin = truncF(minV / step); smax = truncF(maxV / step); break;
        case 2: smin = 0.0f; smax = truncF((maxV - minV) / step); break;
        default: smin = minV; smax = maxV; break;
    }
    const float val = (value < smin) ? smin : ((value > smax) ? smax : value);
    float rawV;
    switch (spinner) {
        case 0: rawV = val * v.mult; break;
        case 1: rawV = (val * step) * v.mult; break;
        case 2: rawV = (val * step + minV) * v.mult; break;
        default: rawV = val; break;
    }
    if (spinner == 4 && !predefined.empty()) {
        int best = 0; float bd = 3.402823466e+38f;
        for (int i = 0; i < (int)predefined.size(); ++i) {
            const float d = fabsf(predefined[i] - rawV);
            if (d < bd) { bd = d; best = i; }
        }
        rawV = predefined[best];
    }
    setRaw(rawV);
    return true;
}

bool setScoringVar(pdb_car_params& P, const std::string& name, float w) {
    pdb_scoring& s = P.scoring;
#define SV(n) if (name == #n) { s.n = w; return true; }
    SV(SmoothSteerSpeed) SV(MinBonusSpeed) SV(MaxBonusSpeed) SV(StallRpm) SV(DirectionThreshold) SV(OutOfTrackThreshold)
    SV(ApproachDistance) SV(CriticalDistance) SV(TravelBonus) SV(TravelSplineBonus) SV(DriftBonus) SV(SpeedBonus)
    SV(ThrottleBonus) SV(EngineRpmBonus) SV(DirectionBonus) SV(DirectionPenalty) SV(ObstApproachPenalty) SV(CollisionPenalty)
    SV(OffTrackPenalty) SV(GearGrindPenalty) SV(StallPenalty)
#undef SV
    return false;
}
bool getScoringVar(const pdb_car_params& P, const std::string& name, float& w) {
    const pdb_scoring& s = P.scoring;
#define SV(n) if (name == #n) { w = s.n; return true; }
    SV(SmoothSteerSpeed) SV(MinBonusSpeed) SV(MaxBonusSpeed) SV(StallRpm) SV(DirectionThreshold) SV(OutOfTrackThreshold)
    SV(ApproachDistance) SV(CriticalDistance) SV(TravelBonus) SV(TravelSplineBonus) SV(DriftBonus) SV(SpeedBonus)
    SV(ThrottleBonus) SV(EngineRpmBonus) SV(DirectionBonus) SV(DirectionPenalty) SV(ObstApproachPenalty) SV(CollisionPenalty)
    SV(OffTrackPenalty) SV(GearGrindPenalty) SV(StallPenalty)
#undef SV
    w = 0;
    return false;
}

}  // namespace pdb
