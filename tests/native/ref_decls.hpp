// Declarations TRANSCRIBED from the reference's headers -- only the members include/isvins_estimator_shim.hpp reads or
// writes, each with the file:line it is declared at (relative to /root/reference) -- so that the shim can be type-checked
// without Eigen / Ceres / Sophus / OpenCV (absent from this image).  TEST INFRASTRUCTURE for
// tests/test_shim_typecheck.py: "type-check only, pins nothing".  No function bodies, no arithmetic.
#pragma once
#include <Eigen/Dense>
#include <iostream>
#include <list>
#include <mutex>
#include <queue>
#include <vector>
using namespace Eigen;      // include/estimator.h:20-21 and the factor headers do the same
using namespace std;

// include/parameters.h:13,35,37,40,14,42,44,51,56,82-87
const int NUM_OF_CAM = 1;
const int Vo_SIZE = 8;
const int NUM_OF_F = 1000;
const int ALL_BUF_SIZE = 18;
extern double ALPHA;
extern double INIT_DEPTH;
extern int ESTIMATE_EXTRINSIC;
extern Eigen::Vector3d G;
extern int NUM_ITERATIONS;
enum SIZE_PARAMETERIZATION { SIZE_POSE = 7, SIZE_SPEEDBIAS = 9, SIZE_FEATURE = 1 };

// include/factor/integration_base.h:188-207
class IntegrationBase {
  public:
    Eigen::Vector3d linearized_ba, linearized_bg;                    // :193
    Eigen::Matrix<double, 15, 15> jacobian, covariance;              // :195
    double sum_dt;                                                   // :200
    Eigen::Vector3d delta_p;                                         // :201
    Eigen::Quaterniond delta_q;                                      // :202
    Eigen::Vector3d delta_v;                                         // :203
};

// include/factor/projection_factor.h:48
class ProjectionFactor {
  public:
    static Eigen::Matrix2d sqrt_info;
};

// include/factor/relative_pose_factor.h:13-24,190-194
class RelativePoseFactor {
  public:
    RelativePoseFactor() = delete;                                   // :16
    RelativePoseFactor(const Vector3d delta_t_, const Matrix3d delta_R_);   // :18
    void setIndex(int i, int j);                                     // :22
    Vector3d delta_t;                                                // :190
    Matrix3d delta_R;                                                // :191
    MatrixXd sqrt_info;                                              // :192
    int imu_i, imu_j;                                                // :194
};
// include/factor/se3_prior_factor.h:9-19,135-138
class SE3PriorFactor {
  public:
    SE3PriorFactor(const Vector3d t_new, const Quaterniond R_new);   // :13
    void setIndex(int i);                                            // :17
    Vector3d t;                                                      // :135
    Matrix3d R;                                                      // :136
    MatrixXd sqrt_info;                                              // :137
    int index;                                                       // :138
};
// include/factor/linear9_factor.h:8-18,69-72   (the second constructor argument, a ceres::Ownership, is defaulted)
class Linear9Factor {
  public:
    Linear9Factor(const Matrix<double, 9, 1> VB_);                   // :12
    void setIndex(int i);                                            // :16
    Matrix<double, 9, 1> VB;                                         // :69
    MatrixXd sqrt_info;                                              // :70
    int index;                                                       // :72
};
// include/factor/rollpitch_factor.h:10-24,132-134
class RollPitchFactor {
  public:
    RollPitchFactor(const Quaterniond Rz);                           // :14
    void setIndex(int i);                                            // :18
    Matrix3d R;                                                      // :132
    MatrixXd sqrt_info;                                              // :133
    int index;                                                       // :134
};
// include/factor/pose_graph_factors.h:6-26
struct CombinedFactors {
    RelativePoseFactor *relativePoseFactor;                          // :7
    RollPitchFactor *rollPitchFactor;                                // :8
    long vio_index;                                                  // :9
    int length;                                                      // :10
    long pg_index;                                                   // :11
    Eigen::MatrixXd covRel;                                          // :12
    Eigen::MatrixXd covAbs;                                          // :13
    double distance;                                                 // :14
    double ts;                                                       // :15
    Eigen::Matrix3d Ri;                                              // :16
    Eigen::Vector3d ti;                                              // :17
    CombinedFactors(long index = 0);                                 // :19
};

// include/feature_tracker/feature_manager.h:21-40,42-63,66-93
class Feature {
  public:
    Vector3d point;                                                  // :35
};
class IDFeatures {
  public:
    int start_frame;                                                 // :46
    vector<Feature> idfeatures;                                      // :47
    int used_num;                                                    // :50
    double estimated_depth;                                          // :52
    int solve_flag;                                                  // :53
};
class FeatureManager {
  public:
    bool goodFeature(IDFeatures &idfs);                              // :75
    list<IDFeatures> IDsfeatures;                                    // :92
};

// include/estimator.h:28,38-159
static long PoseGraphFactorCount = 0;                                // :28  (file-scope static, NOT a member)
class Estimator {
  public:
    enum MarginalizationFlag { MARGIN_OLD = 0, MARGIN_NEW = 1 };     // :78-82
    MarginalizationFlag marginalization_flag;                        // :85
    Matrix3d ric[NUM_OF_CAM];                                        // :87
    Vector3d tic[NUM_OF_CAM];                                        // :88
    Vector3d Ps[ALL_BUF_SIZE];                                       // :90
    Vector3d Vs[ALL_BUF_SIZE];                                       // :91
    Matrix3d Rs[ALL_BUF_SIZE];                                       // :92
    Vector3d Bas[ALL_BUF_SIZE];                                      // :93
    Vector3d Bgs[ALL_BUF_SIZE];                                      // :94
    double Headers[ALL_BUF_SIZE];                                    // :99
    IntegrationBase *pre_integrations[ALL_BUF_SIZE];                 // :101
    FeatureManager f_manager;                                        // :111
    double para_Pose[ALL_BUF_SIZE][SIZE_POSE];                       // :122
    double para_SpeedBias[ALL_BUF_SIZE][SIZE_SPEEDBIAS];             // :123
    double para_Feature[NUM_OF_F][SIZE_FEATURE];                     // :124
    double para_Ex_Pose[NUM_OF_CAM][SIZE_POSE];                      // :125
    std::mutex m_pose_graph_buf;                                     // :131
    std::queue<CombinedFactors *> pose_graph_factors_buf;            // :132
    Linear9Factor *vioVBPrior;                                       // :135
    vector<RelativePoseFactor *> vioRelativePoseEdges;               // :136
    vector<RollPitchFactor *> vioRollPitchEdges;                     // :137
    SE3PriorFactor *vioPosePriorEdge;                                // :138
    vector<int> MargPointIdx;                                        // :142
    vector<IDFeatures> features2Marg;                                // :143
    SE3PriorFactor *forwardPosePriorEdgeToAdd;                       // :147
    Linear9Factor *backwardVBEdgeToAdd;                              // :153
    RelativePoseFactor *backwardRelativePoseEdgeToAdd;               // :154
    // the ONE member a maintainer adds: the backend handle
    struct isv_backend *isv_handle;
};
