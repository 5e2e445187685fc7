from .operator import Operator as Operator
