# C++ declarations of the plugin surface for Cython (include/microscopes/models/{base,distributions,bbnc,dm,noop}.hpp);
# the reference's counterpart is microscopes/_models_h.pxd.  Same class names, so downstream .pyx files that spell
# `shared_ptr[model]`, `hypers`, `group`, `distributions_model[BetaBernoulli]` compile unchanged against these.
from libcpp.memory cimport shared_ptr
from libcpp.string cimport string

from common_amd.cy._runtime_type_h cimport runtime_type

ctypedef string hyperparam_bag_t
ctypedef string suffstats_bag_t

cdef extern from "microscopes/common/random_fwd.hpp" namespace "microscopes::common":
    cdef cppclass rng_t:
        rng_t() except +
        rng_t(unsigned long) except +

cdef extern from "microscopes/common/runtime_value.hpp" namespace "microscopes::common":
    cdef cppclass value_accessor:
        value_accessor() except +
        value_accessor(const unsigned char *, const bint *, const runtime_type &) except +
    cdef cppclass value_mutator:
        value_mutator() except +
        value_mutator(unsigned char *, const runtime_type &) except +

cdef extern from "microscopes/models/base.hpp" namespace "microscopes::models":
    cdef cppclass hypers

    cdef cppclass group:
        void add_value(const hypers &, const value_accessor &, rng_t &) except +
        void remove_value(const hypers &, const value_accessor &, rng_t &) except +
        float score_value(const hypers &, const value_accessor &, rng_t &) except +
        float score_data(const hypers &, rng_t &) except +
        void sample_value(const hypers &, value_mutator &, rng_t &) except +
        suffstats_bag_t get_ss() except +
        void set_ss(const suffstats_bag_t &) except +
        string debug_str() except +

    cdef cppclass hypers:
        hyperparam_bag_t get_hp() except +
        void set_hp(const hyperparam_bag_t &) except +
        shared_ptr[group] create_group(rng_t &) except +
        string debug_str() except +

    cdef cppclass model:
        shared_ptr[hypers] create_hypers() except +
        runtime_type get_runtime_type() except +

    ctypedef group* group_raw_ptr
    ctypedef shared_ptr[group] group_shared_ptr
    ctypedef hypers* hypers_raw_ptr
    ctypedef shared_ptr[hypers] hypers_shared_ptr
    ctypedef model* model_raw_ptr
    ctypedef shared_ptr[model] model_shared_ptr

# the tag types that select a kernel family (upstream they are the `distributions` library's model classes)
cdef extern from "microscopes/models/distributions.hpp" namespace "distributions":
    cdef cppclass BetaBernoulli:
        pass
    cdef cppclass BetaNegativeBinomial:
        pass
    cdef cppclass GammaPoisson:
        pass
    cdef cppclass NormalInverseChiSq:
        pass
    cdef cppclass DirichletDiscrete128:
        pass
    cdef cppclass NormalInverseWishartV:
        pass

cdef extern from "microscopes/models/distributions.hpp" namespace "microscopes::models":
    cdef cppclass distributions_model[T](model):
        distributions_model() except +

    cdef cppclass distributions_model_dd128(model):
        distributions_model_dd128(unsigned) except +

    cdef cppclass distributions_model_niwv(model):
        distributions_model_niwv(unsigned) except +

cdef extern from "microscopes/models/bbnc.hpp" namespace "microscopes::models":
    cdef cppclass bbnc_model(model):
        bbnc_model() except +

cdef extern from "microscopes/models/dm.hpp" namespace "microscopes::models":
    cdef cppclass dm_model(model):
        dm_model(unsigned) except +

cdef extern from "microscopes/models/noop.hpp" namespace "microscopes::models":
    cdef cppclass noop_model(model):
        noop_model() except +
