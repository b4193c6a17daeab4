class ListTensor(object):
    pass
