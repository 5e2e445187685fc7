from .distribute import (
    init_process_group as init_process_group,
    distribute as distribute,
    gather_field as gather_field,
    SlabPlan as SlabPlan,
    barrier as barrier,
    all_reduce_max as all_reduce_max,
)
