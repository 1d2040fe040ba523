// pmmvps_host.hpp -- host-side mirror of the reference's class surface for the propagate+optim path.
//
// Same class names, public fields and call sequence as imkaywu/MVSKit (test/test.cpp:155-161):
//     Option option;  option.init(prefix, "option");
//     PmMvps pmmvps;  pmmvps.init(option);  pmmvps.run();
// but dependency-free (no Eigen / CImg / NLopt) and with Propagate::run forwarding to the MI355X engine
// through the C ABI of include/mvskit_engine.h.  What stays on the host is what the reference keeps
// around the hot path: option parsing (pmmvps/option.cpp), camera/image/patch file I/O
// (image/camera.cpp:27-63, image/photoSet.cpp:20-61, pmmvps/patch.cpp:31-79,
// pmmvps/patch_manager.cpp:435-540) and the iteration loop (pmmvps/pmmvps.cpp:76-114).
#pragma once
#include <array>
#include <iostream>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "mvskit_engine.h"

namespace mvshost {

using std::string;
using std::vector;
typedef std::array<float, 4> Vector4f;

// pmmvps/option.hpp:20-73
struct Option {
    Option();
    // option.hpp:24.  Where the reference prints and exit(1)s (option.cpp:113-128) the message goes to std::cerr and
    // m_status becomes -1: a library must not end the process of its caller.
    void init(const string prefix, const string option);
    int m_status = 0;

    int m_nimages, m_nillums, m_level, m_csize;
    float m_nccThreshold;
    int m_wsize, m_minImageNum, m_cpu, m_setEdge, m_useBound, m_useVisData, m_sequence;
    float m_maxAngleThreshold, m_quadThreshold;
    string m_prefix, m_option;
    int m_flag;
    vector<int> m_images;
    std::map<int, int> m_dict;
    vector<vector<int> > m_visdata, m_visdata2;

protected:
    void initVisdata();
};

// image/photo.hpp + camera.hpp + image.hpp, reduced to what crosses the boundary: level-0 image and projection
struct Photo {
    int m_width = 0, m_height = 0;
    float m_projection[12] = {0};       // Camera::m_projections[0], row-major 3x4
    vector<unsigned char> m_image;      // Image::m_images[0], interleaved RGB
    vector<unsigned char> m_mask;       // Image::m_masks[0] or empty
    int m_txtType = 0;
    float m_R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // Camera::setR (camera.cpp:179-197): the rotation of a CONTOUR2 camera, row-major
    // Camera::init for CONTOUR / CONTOUR2 text files (camera.cpp:27-63,102-141,241-261)
    int initCamera(const string cname);
    // binary PPM (P6) reader; PhotoSet::init accepts `image/%04d%04d.ppm` (photoSet.cpp:33-36)
    int readPpm(const string iname);
    // Image::readJpeg (image.cpp:827-879): grey files become R = G = B; decoder in jpeg_decode.cpp
    int readJpeg(const string iname);
    int readPgmMask(const string mname);  // P5, thresholded at 127 (image.cpp:170-177)
    int readPbmMask(const string mname);  // P4, Image::readPBMImage (image.cpp:881-946)
};

// image/photoSet.hpp:23-62
class PhotoSet {
public:
    // photoSet.hpp:23 (void there too; where the reference exit(1)s on an unreadable camera or image, m_status becomes -1)
    void init(const vector<int>& images, const string prefix, const int nimages, const int nillums, const int maxLevel, const int size, const int alloc);
    int m_status = 0;
    // in-memory injection (synthetic scenes)
    void setPhoto(int index, int width, int height, const float P[12], const unsigned char* rgb, const unsigned char* mask);
    int getWidth(const int index, const int level) const { return m_photos[index].m_width >> level; }
    int getHeight(const int index, const int level) const { return m_photos[index].m_height >> level; }
    void project(const int index, const Vector4f& coord, const int level, float icoord[3]) const;  // photoSet.cpp:243 -> camera.cpp:310-326
    int getMask(const int index, const int ix, const int iy, const int level) const;               // photoSet.cpp:215 -> image.cpp:765-781 (level 0 only here)
    int image2index(const int image) const;
    vector<Photo> m_photos;
    vector<int> m_images;
    int m_nimages = 0, m_nillums = 1;
    string m_prefix;
    std::map<int, int> m_dict;
};

// pmmvps/patch.hpp:23-67
class Patch {
public:
    Patch();
    float score2(const float threshold) const;
    Vector4f m_coord, m_normal;
    vector<int> m_images, m_vimages;
    float m_ncc;
    int m_nimages, m_iter, m_collected, m_flag;
    unsigned char m_dflag;
    int m_fix, m_id;
    float m_dscale, m_ascale, m_tmp;
};
typedef std::shared_ptr<Patch> Ppatch;
std::istream& operator>>(std::istream& istr, Patch& rhs);        // patch.cpp:31-56
std::ostream& operator<<(std::ostream& ostr, const Patch& rhs);  // patch.cpp:58-79

class PmMvps;

// pmmvps/patch_manager.hpp: the grids live on the device; the host keeps m_ppatches and the file formats
class PatchManager {
public:
    explicit PatchManager(PmMvps& pmmvps) : m_pmmvps(pmmvps) {}
    void init();
    void image2index(Patch& patch);
    void index2image(Patch& patch);
    void collectPatches(const int target = 0);  // downloads the pool (patch_manager.cpp:75-105)
    int readPatches();                          // ply/00000000.patch -> engine (patch_manager.cpp:435-466)
    int readPatches(const int iter);
    void addPatches(const vector<Ppatch>& seeds);  // in-memory seeds
    void writePatches(const string prefix, bool bExportPLY, bool bExportPatch, bool bExportPSet);  // :499-540
    void writePly(const vector<Ppatch>& ppatches, const string filename);                           // :542-633, colour = mean of the views' samples
    vector<int> m_gheights, m_gwidths;
    vector<Ppatch> m_ppatches;

protected:
    int upload(const vector<Ppatch>& pp);
    PmMvps& m_pmmvps;
};

// pmmvps/propagate.hpp:29-69
class Propagate {
public:
    explicit Propagate(PmMvps& pmmvps) : m_pmmvps(pmmvps) {}
    void init();
    void run(const int iter);  // propagate.hpp:30; a failure is reported through PmMvps::m_status
    int MAX_NUM_OF_PATCHES = 0, MAX_NUM_OF_PROPAG = 0;
    long long m_ecount = 0, m_fcount0 = 0, m_fcount1 = 0, m_pcount = 0;
    mvs_counters m_counters{};

protected:
    PmMvps& m_pmmvps;
};

// pmmvps/optim.hpp:24-112: the three calls Propagate::propagatePatch makes on a candidate (propagate.cpp:182-196), here on
// one host-side Patch at a time through the engine's batched single-function entry (mvs_engine_probe).  Inside
// Propagate::run they never cross the boundary -- the sweep kernel runs them on the device -- so this class is for callers
// that drive Optim directly, as the reference's other propagation modes and its tests would.
class Optim {
public:
    explicit Optim(PmMvps& pmmvps) : m_pmmvps(pmmvps) {}
    void init() {}
    int preProcess(Patch& patch);                    // optim.cpp:137-163; -1 = rejected
    void refinePatch(Patch& patch, const int time);  // optim.cpp:470-547 (`time` is ignored there too, D12)
    int postProcess(Patch& patch);                   // optim.cpp:260-298; -1 = rejected
    float computeNcc(const Patch& patch);            // PatchManager::computeNcc -> Optim::computeINCC, patch_manager.cpp:401-404

protected:
    int probe(int op, Patch& patch, float* value);
    PmMvps& m_pmmvps;
};

// pmmvps/depth_normal_init.hpp:20-44.  The reference hard-wires isTest = 1 (depth_normal_init.cpp:30): the seeds are
// PatchManager::readPatches() of ply/00000000.patch.  Its other branch (:34-91, readDepths :94-112, readNormals :114-144) --
// seeds from a point cloud ply/00000000.ply and one normal map per view ply/%08d.ply (vertex = pixel x y, normal in camera
// axes) -- runs when m_isTest is set to 0.
class DepthNormInit {
public:
    explicit DepthNormInit(PmMvps& pmmvps) : m_pmmvps(pmmvps) {}
    void init() {}
    void init(const string prefix, const int nfiles) { m_prefix = prefix; m_nplys = nfiles; }  // depth_normal_init.cpp:24-27
    void createPatches();
    // the host part of the PLY branch: the patches createPatches() then hands to PatchManager (no engine involved)
    int buildPatches(vector<Ppatch>& ppatches);
    int m_isTest = 1;

protected:
    int readDepths(vector<std::array<float, 3> >& coords);
    int readNormals(vector<vector<std::array<float, 3> > >& normals);
    void sortImages(Patch& patch, const int isFixed) const;  // Optim::sortImages, optim.cpp:221-258 (+ computeUnits 86-107, getUnit 34-41)
    PmMvps& m_pmmvps;
    string m_prefix;
    int m_nplys = 0;
};

// pmmvps/filter.hpp:24-63
class Filter {
public:
    explicit Filter(PmMvps& pmmvps) : m_pmmvps(pmmvps) {}
    void init() {}
    void run();  // filter.hpp / filter.cpp:25-49, on the engine; a failure is reported through PmMvps::m_status
    long long m_removed[4] = {0, 0, 0, 0};  // filterOutside / filterExact / filterNeighbor / filterSmallGroups

protected:
    PmMvps& m_pmmvps;
};

// pmmvps/pmmvps.hpp:25-107
class PmMvps {
public:
    PmMvps();
    virtual ~PmMvps();
    // pmmvps.hpp:30-31.  Where the reference exit(1)s, m_status becomes a negative mvs_status (0 = fine) and the call
    // returns; later calls do nothing while m_status != 0.
    void init(const Option& option);
    void init(const Option& option, const PhotoSet& photos);  // photos already in memory
    void run();
    int m_status = 0;
    int ITER = 3;  // pmmvps.cpp:90 (compile-time constant there; D13)
    // Multi-GPU (no reference counterpart): one PmMvps per process and GPU.  Call before init(): this process is rank
    // `rank` of `world`; `idFile` is a path all ranks can reach -- rank 0 writes the RCCL communicator id there, the
    // others wait for it (mvs_comm_unique_id / mvs_engine_comm_init).  Every rank then runs the same init()/run(): the
    // engine sweeps its share of the cells and exchanges inside Propagate::run; all ranks end with the same patches.
    void setRanks(int rank, int world, const string& idFile, int device = -1);
    int m_rank = 0, m_world = 1, m_device = 0;
    string m_commIdFile;
    unsigned long long m_jobNonce = 0;  // the same on every rank of a job (0: MVS_JOB_NONCE from the environment, else the id file's age decides)
    long long m_startedNs = 0;          // wall clock when this rank's job began (0: a minute before joinRanks)
    bool m_strictListCap = true;        // init fails when the data set has more views than the linked engine's lists hold (false: a warning, and lists are cut)

    int m_nimages = 0, m_nillums = 1;
    vector<int> m_images;
    string m_prefix;
    int m_level = 1, m_csize = 2;
    float m_nccThreshold = 0.7f;
    int m_wsize = 7, m_minImageNumThreshold = 3;
    vector<vector<int> > m_visdata, m_visdata2;
    float m_quadThreshold = 2.5f;
    int m_tau = 0, m_depth = 0;
    float m_angleThreshold0 = 0, m_angleThreshold1 = 0;
    int m_countThreshold1 = 4;
    float m_neighborThreshold = 0.5f, m_neighborThreshold1 = 1.0f, m_neighborThreshold2 = 1.0f;
    float m_nccThresholdBefore = 0.4f, m_maxAngleThreshold = 0;

    PhotoSet m_photoSet;
    DepthNormInit m_dnInit;
    PatchManager m_patchManager;
    Propagate m_propagate;
    Optim m_optim;
    Filter m_filter;
    mvs_engine* m_engine = nullptr;  // Optim + the PatchManager grids live behind this handle
    unsigned m_seed = 1;
    int m_refineSteps = 6;
    int m_viewPropagation = 0;  // 1 = the branch propagate.cpp:110-120 keeps commented out
    int m_literalGroups = 0;    // 1 = Filter::filterSmallGroups labels breadth-first in patch order, as filter.cpp:432-524 does
    bool m_writeFiles = true;

    void updateThreshold();  // pmmvps.cpp:70-74

protected:
    int createEngine(float maxAngle, float quad);
    int joinRanks();
};

}  // namespace mvshost
